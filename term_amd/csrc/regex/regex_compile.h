// regex_compile.h -- Rust-`regex`-syntax pattern -> byte-level DFA over UTF-8, for the GPU matcher.
//
// The reference evaluates pattern checks as `col ~ 'pat'` in DataFusion, i.e. arrow-string's
// regexp_is_match -> regex::Regex::is_match (regex 1.12.2, Cargo.lock:3637-3638): an UNANCHORED search
// with Unicode-aware \d \w \s, `$` matching only at the end of the haystack, `(?i)` for the `~*` operator
// (TG/constraints/format.rs:756-776).  This front-end parses that syntax, expands classes to UTF-8 byte
// sequences, builds a Thompson NFA and determinises it (the search prefix is folded into the subset
// construction), so the device only walks `state = table[state][class(byte)]`.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace tgx {
namespace rx {

struct Dfa {
  // state 0 = DEAD (no match possible any more), state 1 = MATCHED (absorbing)
  uint32_t n_states = 0;
  uint32_t n_classes = 0;
  uint32_t start = 0;
  uint8_t byte_class[256];
  std::vector<uint16_t> table;         // n_states * n_classes
  std::vector<uint8_t> accept_at_end;  // per state: the haystack may end here with a match
  // A counted repeat of ONE class between anchors -- `^C{m,n}$`: "a user name of 1 to 64 word characters" -- is the
  // automaton of `^C*$` and a count: every character of a matching value is one of C, so the value matches iff the
  // automaton does AND its number of characters lies in [len_min, len_max].  The expanded automaton would hold a copy
  // of the class's UTF-8 form per repetition (\w: ~300 states each).  -1: no bound (every other pattern).
  int64_t len_min = -1, len_max = -1;
};

enum CompileStatus {
  kOk = 0,
  kInvalid = 1,      // regex::Regex::new would reject it ("Invalid regex pattern: ...")
  kUnsupported = 2,  // valid for the reference, outside this engine (caller falls back)
  kRejected = 3      // SqlSecurity::validate_regex_pattern rejects it (length / NUL / ReDoS literals)
};

// SqlSecurity::validate_regex_pattern (TG/security.rs:152-183) without the compile step
CompileStatus validate_pattern_rules(const char *pattern, size_t len, std::string *msg);

CompileStatus compile(const char *pattern, size_t len, bool case_insensitive, Dfa *out,
                      std::string *msg);

// host-side walk of the automaton (used to check compiled patterns on small inputs)
bool dfa_is_match(const Dfa &d, const uint8_t *s, size_t n);

// Several patterns of ONE column in one walk: the product of their automata.  A product state is the tuple of the
// parts' states; the input byte classes are the common refinement of theirs.  States 0 .. n_final - 1 are the tuples
// whose every component is decided (DEAD or MATCHED): absorbing, state index = bit mask of the parts that matched.
// accept_mask[s] has bit k set when part k matches a haystack that ends in state s.  Only reachable tuples are
// built; false when the table would exceed `max_entries` (states x classes) or there are more than 8 parts --
// the caller then walks the parts separately.
struct ProductDfa {
  uint32_t n_parts = 0;
  uint32_t n_states = 0, n_classes = 0, start = 0, n_final = 0;
  uint8_t byte_class[256];
  std::vector<uint16_t> table;       // n_states * n_classes
  std::vector<uint8_t> accept_mask;  // n_states
};
bool dfa_product(const std::vector<const Dfa *> &parts, uint32_t max_entries, ProductDfa *out);
// bit k = part k matches
uint32_t product_match_mask(const ProductDfa &d, const uint8_t *s, size_t n);

}  // namespace rx
}  // namespace tgx
