// regex_compile.cpp -- see regex_compile.h.
#include "regex_compile.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <set>

#include "unicode_tables.h"

namespace tgx {
namespace rx {

namespace {

constexpr uint32_t kMaxCp = 0x10FFFF;
constexpr int kMaxRepeat = 1000;
constexpr size_t kMaxNfaStates = 600000;
constexpr size_t kMaxDfaStates = 60000;  // (state ids are 16 bits)

struct Range {
  uint32_t lo, hi;
};
typedef std::vector<Range> ClassSet;

void normalize(ClassSet &c) {
  std::sort(c.begin(), c.end(), [](const Range &a, const Range &b) { return a.lo < b.lo; });
  ClassSet out;
  for (const Range &r : c) {
    if (!out.empty() && r.lo <= out.back().hi + 1) {
      if (r.hi > out.back().hi) out.back().hi = r.hi;
    } else {
      out.push_back(r);
    }
  }
  c.swap(out);
}

ClassSet negate(ClassSet c) {
  normalize(c);
  ClassSet out;
  uint32_t next = 0;
  for (const Range &r : c) {
    if (r.lo > next) out.push_back({next, r.lo - 1});
    next = r.hi + 1;
  }
  if (next <= kMaxCp) out.push_back({next, kMaxCp});
  return out;
}

// class set operations of regex-syntax: [a&&b] [a--b] [a~~b] (operands are normalised)
ClassSet intersect(const ClassSet &a, const ClassSet &b) {
  ClassSet out;
  size_t i = 0, j = 0;
  while (i < a.size() && j < b.size()) {
    const uint32_t lo = std::max(a[i].lo, b[j].lo), hi = std::min(a[i].hi, b[j].hi);
    if (lo <= hi) out.push_back({lo, hi});
    if (a[i].hi < b[j].hi)
      i++;
    else
      j++;
  }
  return out;
}
ClassSet difference(const ClassSet &a, const ClassSet &b) { return intersect(a, negate(b)); }
ClassSet symmetric_difference(const ClassSet &a, const ClassSet &b) {
  ClassSet u = a;
  u.insert(u.end(), b.begin(), b.end());
  normalize(u);
  return difference(u, intersect(a, b));
}

bool contains(const ClassSet &c, uint32_t cp) {
  size_t lo = 0, hi = c.size();
  while (lo < hi) {
    size_t mid = (lo + hi) / 2;
    if (c[mid].hi < cp)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < c.size() && c[lo].lo <= cp;
}

// simple case folding closure of a class (tgx_fold_pairs: every ordered pair of each orbit)
void case_fold(ClassSet &c) {
  normalize(c);
  ClassSet extra;
  for (uint32_t i = 0; i < tgx_n_fold_pairs; i++)
    if (contains(c, tgx_fold_pairs[i][0])) extra.push_back({tgx_fold_pairs[i][1], tgx_fold_pairs[i][1]});
  c.insert(c.end(), extra.begin(), extra.end());
  normalize(c);
}

bool table_lookup(const std::string &name, ClassSet *out) {
  for (uint32_t i = 0; i < tgx_n_utables; i++) {
    if (name == tgx_utables[i].name) {
      out->clear();
      for (uint32_t k = 0; k < tgx_utables[i].count; k++)
        out->push_back({tgx_utables[i].ranges[k].lo, tgx_utables[i].ranges[k].hi});
      return true;
    }
  }
  return false;
}

// ------------------------------------------------------------------------------------------- AST
struct Node {
  // kLineStart / kLineEnd: `^` / `$` under (?m); kWordB / kNotWordB: ASCII word boundaries (`\b` / `\B` under (?-u));
  // kUWordB / kUNotWordB: the Unicode ones (whole characters on either side, \w as in the tables)
  enum Kind { kEmpty, kClass, kStart, kEnd, kLineStart, kLineEnd, kWordB, kNotWordB, kUWordB, kUNotWordB, kConcat, kAlt,
              kRepeat } kind = kEmpty;
  ClassSet cls;
  std::vector<std::unique_ptr<Node>> kids;
  int min = 0, max = 0;  // kRepeat; max < 0 = unbounded
};
typedef std::unique_ptr<Node> NodeP;

NodeP mk(Node::Kind k) {
  NodeP n(new Node());
  n->kind = k;
  return n;
}

struct Flags {
  bool i = false, s = false, x = false, swap_greed = false;
  bool m = false;   // multi-line: ^ / $ also match after / before a line feed
  bool u = true;    // Unicode mode; (?-u) makes \w \d \s \b ASCII (only forms that cannot match invalid UTF-8 are taken)
  bool crlf = false;
};

struct Parser {
  std::vector<uint32_t> p;  // pattern as code points
  size_t pos = 0;
  CompileStatus status = kOk;
  std::string msg;
  int depth = 0;

  bool fail(CompileStatus st, const std::string &m) {
    if (status == kOk) {
      status = st;
      msg = m;
    }
    return false;
  }
  bool eof() const { return pos >= p.size(); }
  uint32_t peek(size_t k = 0) const { return pos + k < p.size() ? p[pos + k] : 0xFFFFFFFFu; }
  bool eat(uint32_t c) {
    if (peek() == c) {
      pos++;
      return true;
    }
    return false;
  }

  static bool decode_utf8(const char *s, size_t n, std::vector<uint32_t> *out) {
    size_t i = 0;
    while (i < n) {
      uint8_t b = (uint8_t)s[i];
      uint32_t cp;
      int len;
      if (b < 0x80) {
        cp = b;
        len = 1;
      } else if ((b & 0xE0) == 0xC0) {
        cp = b & 0x1F;
        len = 2;
      } else if ((b & 0xF0) == 0xE0) {
        cp = b & 0x0F;
        len = 3;
      } else if ((b & 0xF8) == 0xF0) {
        cp = b & 0x07;
        len = 4;
      } else {
        return false;
      }
      if (i + len > n) return false;
      for (int k = 1; k < len; k++) {
        uint8_t c = (uint8_t)s[i + k];
        if ((c & 0xC0) != 0x80) return false;
        cp = (cp << 6) | (c & 0x3F);
      }
      out->push_back(cp);
      i += len;
    }
    return true;
  }

  void skip_verbose(const Flags &f) {
    if (!f.x) return;
    for (;;) {
      uint32_t c = peek();
      if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == 0x0B) {
        pos++;
      } else if (c == '#') {
        while (!eof() && peek() != '\n') pos++;
      } else {
        break;
      }
    }
  }

  NodeP literal(uint32_t cp, const Flags &f) {
    NodeP n = mk(Node::kClass);
    n->cls.push_back({cp, cp});
    if (f.i) case_fold(n->cls);
    return n;
  }

  // ---- alternation
  NodeP parse_alt(Flags f) {
    if (++depth > 200) {
      fail(kInvalid, "exceeded the maximum nesting depth");
      return nullptr;
    }
    std::vector<NodeP> branches;
    branches.push_back(parse_concat(f));
    if (status != kOk) return nullptr;
    while (peek() == '|') {
      pos++;
      branches.push_back(parse_concat(f));
      if (status != kOk) return nullptr;
    }
    depth--;
    if (branches.size() == 1) return std::move(branches[0]);
    NodeP n = mk(Node::kAlt);
    n->kids = std::move(branches);
    return n;
  }

  // flags set by a bare (?i) group apply to the rest of the enclosing group: `f` is passed by reference
  NodeP parse_concat(Flags &f) {
    NodeP cat = mk(Node::kConcat);
    for (;;) {
      skip_verbose(f);
      uint32_t c = peek();
      if (eof() || c == '|' || c == ')') break;
      NodeP atom = parse_atom(f);
      if (status != kOk) return nullptr;
      if (!atom) continue;  // a flag-only group
      // repetition operators
      for (;;) {
        skip_verbose(f);
        uint32_t q = peek();
        int mn, mx;
        if (q == '*') {
          mn = 0;
          mx = -1;
          pos++;
        } else if (q == '+') {
          mn = 1;
          mx = -1;
          pos++;
        } else if (q == '?') {
          mn = 0;
          mx = 1;
          pos++;
        } else if (q == '{') {
          size_t save = pos;
          if (!parse_counted(&mn, &mx)) {
            if (status != kOk) return nullptr;
            pos = save;
            fail(kInvalid, "repetition quantifier expects a valid decimal");
            return nullptr;
          }
        } else {
          break;
        }
        if (peek() == '?') pos++;  // lazy: irrelevant for is_match
        if (atom->kind == Node::kStart || atom->kind == Node::kEnd) {
          // regex-syntax accepts repetition of assertions (x* of an anchor = optional anchor)
        }
        NodeP rep = mk(Node::kRepeat);
        rep->min = mn;
        rep->max = mx;
        rep->kids.push_back(std::move(atom));
        atom = std::move(rep);
      }
      cat->kids.push_back(std::move(atom));
    }
    if (cat->kids.empty()) return mk(Node::kEmpty);
    if (cat->kids.size() == 1) return std::move(cat->kids[0]);
    return cat;
  }

  bool parse_decimal(int *out) {
    if (!(peek() >= '0' && peek() <= '9')) return false;
    long v = 0;
    while (peek() >= '0' && peek() <= '9') {
      v = v * 10 + (long)(peek() - '0');
      if (v > 100000) {
        fail(kInvalid, "repetition count too large");
        return false;
      }
      pos++;
    }
    *out = (int)v;
    return true;
  }

  bool parse_counted(int *mn, int *mx) {
    pos++;  // {
    while (peek() == ' ') pos++;
    if (!parse_decimal(mn)) return false;
    while (peek() == ' ') pos++;
    if (eat('}')) {
      *mx = *mn;
    } else if (eat(',')) {
      while (peek() == ' ') pos++;
      if (eat('}')) {
        *mx = -1;
      } else {
        if (!parse_decimal(mx)) return false;
        while (peek() == ' ') pos++;
        if (!eat('}')) return false;
        if (*mx < *mn) {
          fail(kInvalid, "invalid repetition count range, the start must be <= the end");
          return false;
        }
      }
    } else {
      return false;
    }
    if (*mn > kMaxRepeat || *mx > kMaxRepeat) {
      fail(kUnsupported, "counted repetition above 1000 is not supported by the GPU engine");
      return false;
    }
    return true;
  }

  NodeP parse_atom(Flags &f) {
    uint32_t c = peek();
    if (c == '(') return parse_group(f);
    if (c == '[') {
      NodeP n = mk(Node::kClass);
      if (!parse_class(f, &n->cls)) return nullptr;
      return n;
    }
    if (c == '.') {
      pos++;
      if (!f.u) {
        fail(kUnsupported, "`.` under (?-u) can match invalid UTF-8; not supported by the GPU engine");
        return nullptr;
      }
      NodeP n = mk(Node::kClass);
      if (f.s)
        n->cls.push_back({0, kMaxCp});
      else
        n->cls = negate({{'\n', '\n'}});
      return n;
    }
    if (c == '^') {
      pos++;
      return mk(f.m ? Node::kLineStart : Node::kStart);
    }
    if (c == '$') {
      pos++;
      return mk(f.m ? Node::kLineEnd : Node::kEnd);
    }
    if (c == '*' || c == '+' || c == '?') {
      fail(kInvalid, "repetition operator missing expression");
      return nullptr;
    }
    if (c == '{') {
      fail(kInvalid, "repetition operator missing expression");
      return nullptr;
    }
    if (c == '\\') return parse_escape_atom(f);
    pos++;
    return literal(c, f);
  }

  NodeP parse_group(Flags &f) {
    pos++;  // (
    Flags inner = f;
    if (peek() == '?') {
      // (?:...)  (?P<name>...)  (?<name>...)  (?flags)  (?flags:...)
      if (peek(1) == 'P' && peek(2) == '<') {
        pos += 3;
        if (!skip_group_name()) return nullptr;
      } else if (peek(1) == '<' && peek(2) != '=' && peek(2) != '!') {
        pos += 2;
        if (!skip_group_name()) return nullptr;
      } else if (peek(1) == '=' || peek(1) == '!' || (peek(1) == '<' && (peek(2) == '=' || peek(2) == '!'))) {
        fail(kInvalid, "look-around, including look-ahead and look-behind, is not supported");
        return nullptr;
      } else if (peek(1) == 'P' && (peek(2) == '=' || peek(2) == '>')) {
        fail(kInvalid, "backreferences are not supported");
        return nullptr;
      } else {
        pos++;  // ?
        bool neg = false, any = false;
        for (;;) {
          uint32_t c = peek();
          if (c == ':' || c == ')') break;
          if (c == '-') {
            if (neg) {
              fail(kInvalid, "dangling flag negation operator");
              return nullptr;
            }
            neg = true;
            pos++;
            continue;
          }
          any = true;
          bool on = !neg;
          switch (c) {
            case 'i': inner.i = on; break;
            case 's': inner.s = on; break;
            case 'x': inner.x = on; break;
            case 'U': inner.swap_greed = on; break;
            case 'R': inner.crlf = on; break;  // CRLF mode only changes multi-line anchors
            case 'm': inner.m = on; break;
            case 'u': inner.u = on; break;
            default:
              fail(kInvalid, "unrecognized flag");
              return nullptr;
          }
          pos++;
        }
        if (!any && !neg && peek() == ')') {
          fail(kInvalid, "missing flags");  // "(?)"
          return nullptr;
        }
        if (inner.m && inner.crlf) {
          fail(kUnsupported, "CRLF multi-line mode (?mR) is not supported by the GPU engine");
          return nullptr;
        }
        if (eat(')')) {
          f = inner;  // applies to the rest of the enclosing group
          return nullptr;
        }
        pos++;  // ':'
      }
    }
    NodeP body = parse_alt(inner);
    if (status != kOk) return nullptr;
    if (!eat(')')) {
      fail(kInvalid, "unclosed group");
      return nullptr;
    }
    return body;
  }

  bool skip_group_name() {
    size_t n = 0;
    while (!eof() && peek() != '>') {
      uint32_t c = peek();
      bool ok = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_' || c == '.' || c == '[' || c == ']' ||
                (n > 0 && c >= '0' && c <= '9') || c > 0x7F;
      if (!ok) return fail(kInvalid, "invalid capture group character");
      pos++;
      n++;
    }
    if (n == 0 || !eat('>')) return fail(kInvalid, "empty or unclosed capture group name");
    return true;
  }

  static int hexval(uint32_t c) {
    if (c >= '0' && c <= '9') return (int)(c - '0');
    if (c >= 'a' && c <= 'f') return (int)(c - 'a' + 10);
    if (c >= 'A' && c <= 'F') return (int)(c - 'A' + 10);
    return -1;
  }

  bool parse_hex(int digits, uint32_t *out) {
    // \xNN, \uNNNN, \UNNNNNNNN or the braced form \x{N...}
    uint32_t v = 0;
    if (peek() == '{') {
      pos++;
      int n = 0;
      while (peek() != '}') {
        int h = hexval(peek());
        if (h < 0 || n >= 8) return fail(kInvalid, "invalid hexadecimal escape");
        v = v * 16 + (uint32_t)h;
        n++;
        pos++;
      }
      pos++;
      if (n == 0) return fail(kInvalid, "empty hexadecimal escape");
    } else {
      for (int k = 0; k < digits; k++) {
        int h = hexval(peek());
        if (h < 0) return fail(kInvalid, "invalid hexadecimal escape");
        v = v * 16 + (uint32_t)h;
        pos++;
      }
    }
    if (v > kMaxCp || (v >= 0xD800 && v <= 0xDFFF)) return fail(kInvalid, "hexadecimal escape is not a Unicode scalar value");
    *out = v;
    return true;
  }

  // \d \w \s \p{..} and negations; returns false (without error) when `c` is not a class escape
  bool class_escape(uint32_t c, const Flags &f, ClassSet *out, bool *is_class) {
    *is_class = true;
    if (!f.u) {  // (?-u): the ASCII Perl classes; their negations could match invalid UTF-8
      switch (c) {
        case 'd': out->assign(1, Range{'0', '9'}); return true;
        case 's': *out = {{'\t', '\r'}, {' ', ' '}}; return true;
        case 'w': *out = {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}}; return true;
        case 'D': case 'S': case 'W': case 'p': case 'P':
          return fail(kUnsupported, "negated / Unicode classes under (?-u) are not supported by the GPU engine");
        default: break;
      }
    }
    switch (c) {
      case 'd': table_lookup("perl_digit", out); return true;
      case 'D': table_lookup("perl_digit", out); *out = negate(*out); return true;
      case 's': table_lookup("perl_space", out); return true;
      case 'S': table_lookup("perl_space", out); *out = negate(*out); return true;
      case 'w': table_lookup("perl_word", out); return true;
      case 'W': table_lookup("perl_word", out); *out = negate(*out); return true;
      case 'p':
      case 'P': {
        std::string name;
        bool neg = c == 'P';
        if (peek() == '{') {
          pos++;
          if (peek() == '^') {
            neg = !neg;
            pos++;
          }
          while (!eof() && peek() != '}') {
            name.push_back((char)peek());
            pos++;
          }
          if (!eat('}')) return fail(kInvalid, "unclosed Unicode class");
        } else {
          if (eof()) return fail(kInvalid, "incomplete escape sequence");
          name.push_back((char)peek());
          pos++;
        }
        if (!unicode_property(name, out)) return false;
        if (neg) *out = negate(*out);
        (void)f;
        return true;
      }
      default:
        *is_class = false;
        return true;
    }
  }

  bool unicode_property(std::string name, ClassSet *out) {
    // General_Category / a few binary properties and scripts; name matching is loose like UAX44-LM3
    std::string key;
    for (char ch : name)
      if (ch != ' ' && ch != '_' && ch != '-') key.push_back((char)tolower((unsigned char)ch));
    size_t eq = key.find('=');
    std::string lhs = eq == std::string::npos ? "" : key.substr(0, eq);
    if (eq != std::string::npos) key = key.substr(eq + 1);
    static const char *const gc_long[][2] = {
        {"letter", "L"}, {"uppercaseletter", "Lu"}, {"lowercaseletter", "Ll"}, {"titlecaseletter", "Lt"},
        {"modifierletter", "Lm"}, {"otherletter", "Lo"}, {"mark", "M"}, {"nonspacingmark", "Mn"},
        {"spacingmark", "Mc"}, {"enclosingmark", "Me"}, {"number", "N"}, {"decimalnumber", "Nd"},
        {"letternumber", "Nl"}, {"othernumber", "No"}, {"punctuation", "P"}, {"connectorpunctuation", "Pc"},
        {"dashpunctuation", "Pd"}, {"openpunctuation", "Ps"}, {"closepunctuation", "Pe"},
        {"initialpunctuation", "Pi"}, {"finalpunctuation", "Pf"}, {"otherpunctuation", "Po"}, {"symbol", "S"},
        {"mathsymbol", "Sm"}, {"currencysymbol", "Sc"}, {"modifiersymbol", "Sk"}, {"othersymbol", "So"},
        {"separator", "Z"}, {"spaceseparator", "Zs"}, {"lineseparator", "Zl"}, {"paragraphseparator", "Zp"},
        {"other", "C"}, {"control", "Cc"}, {"format", "Cf"}, {"privateuse", "Co"}, {"unassigned", "Cn"},
        {"digit", "Nd"}, {"punct", "P"}};
    for (auto &g : gc_long)
      if (key == g[0]) key = g[1];
    std::string lower = key;
    for (auto &ch : lower) ch = (char)tolower((unsigned char)ch);
    for (uint32_t i = 0; i < tgx_n_utables; i++) {
      std::string t = tgx_utables[i].name;  // gc_Lu, prop_Alphabetic, script_Greek
      size_t us = t.find('_');
      std::string kind = t.substr(0, us), val = t.substr(us + 1);
      if (kind == "perl") continue;
      std::string v2;
      for (char ch : val)
        if (ch != '_') v2.push_back((char)tolower((unsigned char)ch));
      if (v2 != lower) continue;
      if (!lhs.empty()) {
        bool ok = (kind == "gc" && (lhs == "gc" || lhs == "generalcategory")) ||
                  (kind == "script" && (lhs == "sc" || lhs == "script"));
        if (!ok) continue;
      }
      out->clear();
      for (uint32_t k = 0; k < tgx_utables[i].count; k++)
        out->push_back({tgx_utables[i].ranges[k].lo, tgx_utables[i].ranges[k].hi});
      return true;
    }
    if (lower == "any") {
      out->assign(1, Range{0, kMaxCp});
      return true;
    }
    if (lower == "ascii") {
      out->assign(1, Range{0, 0x7F});
      return true;
    }
    return fail(kUnsupported, "Unicode property \\p{" + name + "} is not in the GPU engine's tables");
  }

  // single-code-point escapes shared by atoms and classes; *handled=false if `c` is not one
  bool simple_escape(uint32_t c, uint32_t *cp, bool *handled) {
    *handled = true;
    switch (c) {
      case 'n': *cp = '\n'; return true;
      case 't': *cp = '\t'; return true;
      case 'r': *cp = '\r'; return true;
      case 'f': *cp = '\f'; return true;
      case 'v': *cp = 0x0B; return true;
      case 'a': *cp = 0x07; return true;
      case '0': *cp = 0; return fail(kInvalid, "octal escapes are not supported");
      case 'x': return parse_hex(2, cp);
      case 'u': return parse_hex(4, cp);
      case 'U': return parse_hex(8, cp);
      default: break;
    }
    // regex-syntax: any ASCII punctuation may be escaped; letters/digits without a meaning may not
    bool punct = c < 0x80 && !((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z')) && c > ' ';
    if (punct || c == ' ') {
      *cp = c;
      return true;
    }
    *handled = false;
    return true;
  }

  NodeP parse_escape_atom(const Flags &f) {
    pos++;  // backslash
    if (eof()) {
      fail(kInvalid, "incomplete escape sequence, reached end of pattern prematurely");
      return nullptr;
    }
    uint32_t c = peek();
    pos++;
    if (c == 'A') return mk(Node::kStart);
    if (c == 'z') return mk(Node::kEnd);
    if (c == 'b' || c == 'B') {
      // The ASCII ones of (?-u) look at one byte each way (a one-byte context per state); the Unicode ones at whole
      // characters: the automaton's context then follows the bytes of a character through a classifier of \w, and a
      // thread that passed the assertion carries what the NEXT character has to be until that character is through.
      if (peek() == '{') {
        fail(kUnsupported, "word boundary variants \\b{...} are not supported by the GPU engine");
        return nullptr;
      }
      if (f.u) return mk(c == 'b' ? Node::kUWordB : Node::kUNotWordB);
      return mk(c == 'b' ? Node::kWordB : Node::kNotWordB);
    }
    if (c >= '1' && c <= '9') {
      fail(kInvalid, "backreferences are not supported");
      return nullptr;
    }
    ClassSet cs;
    bool is_class = false;
    if (!class_escape(c, f, &cs, &is_class)) return nullptr;
    if (is_class) {
      NodeP n = mk(Node::kClass);
      n->cls = cs;
      if (f.i) case_fold(n->cls);
      return n;
    }
    uint32_t cp = 0;
    bool handled = false;
    if (!simple_escape(c, &cp, &handled)) return nullptr;
    if (!handled) {
      fail(kInvalid, "unrecognized escape sequence");
      return nullptr;
    }
    return literal(cp, f);
  }

  bool posix_class(const std::string &name, ClassSet *out) {
    struct P {
      const char *n;
      ClassSet s;
    };
    static const P tbl[] = {
        {"alnum", {{'0', '9'}, {'A', 'Z'}, {'a', 'z'}}},
        {"alpha", {{'A', 'Z'}, {'a', 'z'}}},
        {"ascii", {{0, 0x7F}}},
        {"blank", {{'\t', '\t'}, {' ', ' '}}},
        {"cntrl", {{0, 0x1F}, {0x7F, 0x7F}}},
        {"digit", {{'0', '9'}}},
        {"graph", {{'!', '~'}}},
        {"lower", {{'a', 'z'}}},
        {"print", {{' ', '~'}}},
        {"punct", {{'!', '/'}, {':', '@'}, {'[', '`'}, {'{', '~'}}},
        {"space", {{'\t', '\r'}, {' ', ' '}}},
        {"upper", {{'A', 'Z'}}},
        {"word", {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}}},
        {"xdigit", {{'0', '9'}, {'A', 'F'}, {'a', 'f'}}},
    };
    for (auto &e : tbl)
      if (name == e.n) {
        *out = e.s;
        return true;
      }
    return false;
  }

  // [items] with regex-syntax's set operations: `&&` (intersection), `--` (difference), `~~` (symmetric difference)
  // bind weaker than the union of adjacent items and associate to the left: [a-z&&[^m]--x] = (([a-z] && [^m]) -- [x])
  bool parse_class(const Flags &f, ClassSet *out) {
    pos++;  // [
    bool neg = eat('^');
    if (neg && !f.u) return fail(kUnsupported, "negated classes under (?-u) are not supported by the GPU engine");
    ClassSet acc, result;
    int pending_op = 0;  // 0 none yet, '&', '-', '~'
    bool first = true;
    auto finish_operand = [&]() {
      normalize(acc);
      if (f.i) case_fold(acc);
      if (pending_op == 0)
        result = acc;
      else if (pending_op == '&')
        result = intersect(result, acc);
      else if (pending_op == '-')
        result = difference(result, acc);
      else
        result = symmetric_difference(result, acc);
      acc.clear();
    };
    for (;;) {
      if (eof()) return fail(kInvalid, "unclosed character class");
      uint32_t c = peek();
      if (c == ']' && !first) {
        pos++;
        break;
      }
      first = false;
      if (c == '[') {
        if (peek(1) == ':') {
          // [:name:] / [:^name:]
          size_t save = pos;
          pos += 2;
          bool pneg = eat('^');
          std::string name;
          while (!eof() && peek() != ':' && name.size() < 16) {
            name.push_back((char)peek());
            pos++;
          }
          ClassSet pc;
          if (peek() == ':' && peek(1) == ']' && posix_class(name, &pc)) {
            pos += 2;
            if (pneg) {
              if (!f.u) return fail(kUnsupported, "negated classes under (?-u) are not supported by the GPU engine");
              pc = negate(pc);
            }
            acc.insert(acc.end(), pc.begin(), pc.end());
            continue;
          }
          pos = save;
        }
        ClassSet nested;
        if (!parse_class(f, &nested)) return false;
        acc.insert(acc.end(), nested.begin(), nested.end());
        continue;
      }
      if ((c == '&' && peek(1) == '&') || (c == '-' && peek(1) == '-') || (c == '~' && peek(1) == '~')) {
        finish_operand();
        pending_op = (int)c;
        pos += 2;
        continue;
      }
      uint32_t lo;
      bool lo_is_class = false;
      ClassSet cs;
      if (!class_item(f, &lo, &cs, &lo_is_class)) return false;
      if (lo_is_class) {
        acc.insert(acc.end(), cs.begin(), cs.end());
        continue;
      }
      if (peek() == '-' && peek(1) != ']' && peek(1) != '-' && !eof()) {
        pos++;  // -
        uint32_t hi;
        bool hi_is_class = false;
        ClassSet hs;
        if (!class_item(f, &hi, &hs, &hi_is_class)) return false;
        if (hi_is_class) return fail(kInvalid, "invalid character class range, a class is not a valid range endpoint");
        if (hi < lo) return fail(kInvalid, "invalid character class range, the start must be <= the end");
        acc.push_back({lo, hi});
      } else {
        acc.push_back({lo, lo});
      }
    }
    finish_operand();
    if (neg) result = negate(result);
    *out = result;
    return true;
  }

  bool class_item(const Flags &f, uint32_t *cp, ClassSet *cs, bool *is_class) {
    *is_class = false;
    uint32_t c = peek();
    pos++;
    if (c != '\\') {
      *cp = c;
      return true;
    }
    if (eof()) return fail(kInvalid, "incomplete escape sequence, reached end of pattern prematurely");
    uint32_t e = peek();
    pos++;
    if (e == 'b') {
      *cp = 0x08;  // inside a class \b is backspace in regex-syntax
      return true;
    }
    if (!class_escape(e, f, cs, is_class)) return false;
    if (*is_class) return true;
    bool handled = false;
    if (!simple_escape(e, cp, &handled)) return false;
    if (!handled) return fail(kInvalid, "unrecognized escape sequence");
    return true;
  }
};

// ------------------------------------------------------------------------------------------- NFA
struct NState {
  enum Type { kByte, kSplit, kEmpty, kAssertStart, kAssertEnd, kAssertLineStart, kAssertLineEnd, kAssertWordB,
              kAssertNotWordB, kAssertUWordB, kAssertUNotWordB, kMatch } type;
  uint8_t lo = 0, hi = 0;
  int out = -1, out1 = -1;
};

struct ByteSeq {
  int n;
  uint8_t lo[4], hi[4];
};

void push_seq(std::vector<ByteSeq> *out, uint32_t s, uint32_t e, int n) {
  ByteSeq q;
  q.n = n;
  if (n == 1) {
    q.lo[0] = (uint8_t)s;
    q.hi[0] = (uint8_t)e;
  } else {
    static const uint8_t lead[5] = {0, 0, 0xC0, 0xE0, 0xF0};
    for (int i = 0; i < n; i++) {
      int shift = 6 * (n - 1 - i);
      uint8_t a = (uint8_t)((s >> shift) & 0x3F), b = (uint8_t)((e >> shift) & 0x3F);
      if (i == 0) {
        a = (uint8_t)(lead[n] | (s >> shift));
        b = (uint8_t)(lead[n] | (e >> shift));
      } else {
        a |= 0x80;
        b |= 0x80;
      }
      q.lo[i] = a;
      q.hi[i] = b;
    }
  }
  out->push_back(q);
}

// split [s, e] (same UTF-8 length n) so every byte position ranges independently (utf8-ranges algorithm)
void split_same_len(uint32_t s, uint32_t e, int n, std::vector<ByteSeq> *out) {
  for (int i = 1; i < n; i++) {
    uint32_t m = (1u << (6 * i)) - 1;
    if ((s & ~m) != (e & ~m)) {
      if ((s & m) != 0) {
        split_same_len(s, s | m, n, out);
        split_same_len((s | m) + 1, e, n, out);
        return;
      }
      if ((e & m) != m) {
        split_same_len(s, (e & ~m) - 1, n, out);
        split_same_len(e & ~m, e, n, out);
        return;
      }
    }
  }
  push_seq(out, s, e, n);
}

void utf8_sequences(const ClassSet &cls, std::vector<ByteSeq> *out) {
  static const uint32_t bounds[4][2] = {{0, 0x7F}, {0x80, 0x7FF}, {0x800, 0xFFFF}, {0x10000, 0x10FFFF}};
  for (const Range &r : cls) {
    // drop surrogates
    Range parts[2];
    int np = 0;
    if (r.hi < 0xD800 || r.lo > 0xDFFF) {
      parts[np++] = r;
    } else {
      if (r.lo < 0xD800) parts[np++] = {r.lo, 0xD7FF};
      if (r.hi > 0xDFFF) parts[np++] = {0xE000, r.hi};
    }
    for (int k = 0; k < np; k++)
      for (int n = 1; n <= 4; n++) {
        uint32_t lo = std::max(parts[k].lo, bounds[n - 1][0]), hi = std::min(parts[k].hi, bounds[n - 1][1]);
        if (lo <= hi) split_same_len(lo, hi, n, out);
      }
  }
}

struct Nfa {
  std::vector<NState> st;
  bool overflow = false;
  int add(NState::Type t, int out = -1, int out1 = -1, uint8_t lo = 0, uint8_t hi = 0) {
    if (st.size() >= kMaxNfaStates) {
      overflow = true;
      return 0;
    }
    NState s;
    s.type = t;
    s.out = out;
    s.out1 = out1;
    s.lo = lo;
    s.hi = hi;
    st.push_back(s);
    return (int)st.size() - 1;
  }

  // A class as a MINIMAL acyclic byte automaton: the trie of its UTF-8 sequences with equal sub-tries shared (nodes
  // are hash-consed bottom-up; neighbouring byte ranges that lead to the same node are one edge).  \w is ~2 000
  // trie states and a few dozen nodes.  A counted repeat instantiates the class once per copy: what the subset
  // construction sees per copy is the NODES, which is what lets `^\w{1,64}$` or `^[\p{L}\p{M}\s'-]{1,100}$` through
  // the state limit (round 6; before, every copy brought the whole trie).
  struct DagEdge {
    uint8_t lo, hi;
    int child;  // node id; -1 = the class is complete
    bool operator<(const DagEdge &o) const {
      if (lo != o.lo) return lo < o.lo;
      if (hi != o.hi) return hi < o.hi;
      return child < o.child;
    }
  };
  struct ClassDag {
    std::vector<std::vector<DagEdge>> nodes;  // children before parents
    int root = -1;                            // -1: the empty class
  };
  std::map<std::vector<std::pair<uint32_t, uint32_t>>, ClassDag> dag_cache;

  static int dag_node(ClassDag &dag, std::map<std::vector<DagEdge>, int> &memo, std::vector<DagEdge> edges) {
    std::sort(edges.begin(), edges.end());
    std::vector<DagEdge> merged;
    for (const DagEdge &e : edges) {
      if (!merged.empty() && merged.back().child == e.child && (int)merged.back().hi + 1 == (int)e.lo)
        merged.back().hi = e.hi;
      else
        merged.push_back(e);
    }
    auto it = memo.find(merged);
    if (it != memo.end()) return it->second;
    dag.nodes.push_back(merged);
    memo[merged] = (int)dag.nodes.size() - 1;
    return (int)dag.nodes.size() - 1;
  }
  // seqs[a..b) share their first `depth` byte ranges; group by the range at `depth`
  static int dag_build(ClassDag &dag, std::map<std::vector<DagEdge>, int> &memo, const std::vector<ByteSeq> &seqs, size_t a,
                       size_t b, int depth) {
    std::vector<DagEdge> edges;
    size_t i = a;
    while (i < b) {
      size_t j = i + 1;
      while (j < b && seqs[j].lo[depth] == seqs[i].lo[depth] && seqs[j].hi[depth] == seqs[i].hi[depth] &&
             seqs[j].n == seqs[i].n)
        j++;
      const int child = seqs[i].n == depth + 1 ? -1 : dag_build(dag, memo, seqs, i, j, depth + 1);
      edges.push_back({seqs[i].lo[depth], seqs[i].hi[depth], child});
      i = j;
    }
    return dag_node(dag, memo, std::move(edges));
  }
  const ClassDag &class_dag(const ClassSet &cls) {
    std::vector<std::pair<uint32_t, uint32_t>> key;
    for (const Range &r : cls) key.emplace_back(r.lo, r.hi);
    auto it = dag_cache.find(key);
    if (it != dag_cache.end()) return it->second;
    ClassDag dag;
    std::vector<ByteSeq> seqs;
    utf8_sequences(cls, &seqs);
    if (!seqs.empty()) {
      std::sort(seqs.begin(), seqs.end(), [](const ByteSeq &x, const ByteSeq &y) {
        if (x.n != y.n) return x.n < y.n;
        for (int i = 0; i < x.n; i++) {
          if (x.lo[i] != y.lo[i]) return x.lo[i] < y.lo[i];
          if (x.hi[i] != y.hi[i]) return x.hi[i] < y.hi[i];
        }
        return false;
      });
      std::map<std::vector<DagEdge>, int> memo;
      // (sequences of different lengths never share a first byte range: one root over all of them)
      dag.root = dag_build(dag, memo, seqs, 0, seqs.size(), 0);
    }
    return dag_cache.emplace(std::move(key), std::move(dag)).first->second;
  }
  // one copy of the class in front of `next`
  int instantiate(const ClassDag &dag, int next) {
    if (dag.root < 0) return add(NState::kByte, next, -1, 1, 0);  // empty class: matches nothing
    std::vector<int> entry(dag.nodes.size(), -1);
    for (size_t k = 0; k < dag.nodes.size(); k++) {  // children come first
      const std::vector<DagEdge> &edges = dag.nodes[k];
      int alt = -1;
      for (size_t e = edges.size(); e-- > 0;) {
        const int target = edges[e].child < 0 ? next : entry[(size_t)edges[e].child];
        const int head = add(NState::kByte, target, -1, edges[e].lo, edges[e].hi);
        alt = alt < 0 ? head : add(NState::kSplit, head, alt);
      }
      entry[k] = alt;
    }
    return entry[(size_t)dag.root];
  }

  int compile(const Node *n, int next) {
    if (overflow) return next;
    switch (n->kind) {
      case Node::kEmpty:
        return next;
      case Node::kStart:
        return add(NState::kAssertStart, next);
      case Node::kEnd:
        return add(NState::kAssertEnd, next);
      case Node::kLineStart:
        return add(NState::kAssertLineStart, next);
      case Node::kLineEnd:
        return add(NState::kAssertLineEnd, next);
      case Node::kWordB:
        return add(NState::kAssertWordB, next);
      case Node::kNotWordB:
        return add(NState::kAssertNotWordB, next);
      case Node::kUWordB:
        return add(NState::kAssertUWordB, next);
      case Node::kUNotWordB:
        return add(NState::kAssertUNotWordB, next);
      case Node::kClass:
        return instantiate(class_dag(n->cls), next);
      case Node::kConcat: {
        int cur = next;
        for (size_t i = n->kids.size(); i-- > 0;) cur = compile(n->kids[i].get(), cur);
        return cur;
      }
      case Node::kAlt: {
        std::vector<int> heads;
        for (auto &k : n->kids) heads.push_back(compile(k.get(), next));
        int alt = heads.back();
        for (size_t k = heads.size() - 1; k-- > 0;) alt = add(NState::kSplit, heads[k], alt);
        return alt;
      }
      case Node::kRepeat: {
        const Node *body = n->kids[0].get();
        int cur = next;
        if (n->max < 0) {
          // body* : loop = Split(body -> loop, next)
          int loop = add(NState::kSplit, -1, next);
          int b = compile(body, loop);
          st[loop].out = b;
          cur = loop;
        } else {
          for (int k = n->min; k < n->max; k++) {
            int b = compile(body, cur);
            cur = add(NState::kSplit, b, next);
          }
        }
        for (int k = 0; k < n->min; k++) cur = compile(body, cur);
        return cur;
      }
    }
    return next;
  }
};

// ------------------------------------------------------------------------------------------- DFA
struct Builder {
  const Nfa &nfa;
  explicit Builder(const Nfa &n) : nfa(n) {}
  std::vector<int> stack;
  std::vector<uint32_t> mark;
  uint32_t epoch = 0;

  // What a position knows about its neighbourhood.  `prev`: the byte before it -- kCtxStart (none: start of the
  // haystack), kCtxNewline, kCtxWord ([0-9A-Za-z_]) or kCtxOther, as finely as the pattern's assertions need
  // (ctx_of).  `next`: the byte after it when known (>= 0), kNextUnknown while the automaton has not consumed it yet,
  // kNextEnd at the end of the haystack.
  enum { kCtxStart = 0, kCtxNewline = 1, kCtxWord = 2, kCtxOther = 3 };
  enum { kNextUnknown = -1, kNextEnd = -2 };
  static bool is_word_byte(int b) {
    return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || b == '_' || (b >= 'a' && b <= 'z');
  }

  // A set element is (NFA state, obligation): e = state + obligation * nfa.st.size().  A thread that has passed a
  // Unicode word boundary carries what the character that FOLLOWS the assertion has to be -- kNeedWord / kNeedOther --
  // while it goes on consuming that very character; the transition that completes the character (the classifier says
  // which kind it was) drops the threads whose obligation fails and clears the others'.  Patterns without such
  // assertions never leave obligation 0, and their elements are the state numbers they always were.
  enum { kNoObligation = 0, kNeedWord = 1, kNeedOther = 2 };
  int enc(int state, int obligation) const { return state + obligation * (int)nfa.st.size(); }
  int state_of(int e) const { return e % (int)nfa.st.size(); }
  int obligation_of(int e) const { return e / (int)nfa.st.size(); }

  // epsilon closure of `seeds`.  Assertions that look BEHIND (\A, (?m)^) are decided from `prev`; those that look
  // AHEAD ($, (?m)$, \b, \B) are decided when `next` is known and otherwise stay in the set as pending states -- the
  // transition on the next byte (or the end of the haystack) runs the closure over the set once more with it.
  // `prev` >= 4: the position lies inside a character (Unicode word boundaries only): no assertion holds there.
  void closure(const std::vector<int> &seeds, int prev, int next, std::vector<int> *out) {
    if (mark.size() != 3 * nfa.st.size()) mark.assign(3 * nfa.st.size(), 0);
    epoch++;
    stack.assign(seeds.begin(), seeds.end());
    while (!stack.empty()) {
      int e = stack.back();
      stack.pop_back();
      if (e < 0) continue;
      int s = state_of(e), ob = obligation_of(e);
      if (next == kNextEnd && ob != kNoObligation) {  // nothing follows: "not a word character"
        if (ob == kNeedWord) continue;
        ob = kNoObligation;
        e = enc(s, ob);
      }
      if (mark[e] == epoch) continue;
      mark[e] = epoch;
      const NState &n = nfa.st[s];
      auto push = [&](int to) {
        if (to >= 0) stack.push_back(enc(to, ob));
      };
      switch (n.type) {
        case NState::kEmpty:
          push(n.out);
          break;
        case NState::kSplit:
          push(n.out);
          push(n.out1);
          break;
        case NState::kAssertStart:
          if (prev == kCtxStart) push(n.out);
          break;
        case NState::kAssertLineStart:
          if (prev == kCtxStart || prev == kCtxNewline) push(n.out);
          break;
        case NState::kAssertEnd:
          if (next == kNextEnd)
            push(n.out);
          else if (next == kNextUnknown)
            out->push_back(e);
          break;
        case NState::kAssertLineEnd:
          if (next == kNextEnd || next == '\n')
            push(n.out);
          else if (next == kNextUnknown)
            out->push_back(e);
          break;
        case NState::kAssertWordB:
        case NState::kAssertNotWordB:
          if (next == kNextUnknown) {
            out->push_back(e);
          } else {
            const bool boundary = (prev == kCtxWord) != (next >= 0 && is_word_byte(next));
            if (boundary == (n.type == NState::kAssertWordB)) push(n.out);
          }
          break;
        case NState::kAssertUWordB:
        case NState::kAssertUNotWordB: {
          if (prev >= 4) break;  // (inside a character: cannot be reached on valid UTF-8)
          const bool prev_word = prev == kCtxWord;
          const bool next_must_be_word = (n.type == NState::kAssertUWordB) ? !prev_word : prev_word;
          if (next == kNextEnd) {
            if (!next_must_be_word) push(n.out);
          } else {
            const int need = next_must_be_word ? kNeedWord : kNeedOther;
            if (ob == kNoObligation || ob == need) {
              if (n.out >= 0) stack.push_back(enc(n.out, need));
            }  // (else: two assertions at one position that contradict each other)
          }
          break;
        }
        case NState::kByte:
        case NState::kMatch:
          out->push_back(e);
          break;
      }
    }
    std::sort(out->begin(), out->end());
    out->erase(std::unique(out->begin(), out->end()), out->end());
  }

  bool has_match(const std::vector<int> &set) const {
    for (int e : set)
      if (e >= 0 && nfa.st[state_of(e)].type == NState::kMatch && obligation_of(e) == kNoObligation) return true;
    return false;
  }
};

// What kind of character the bytes since the last character boundary make: a deterministic walk over the UTF-8 forms of
// \w (tables: perl_word) and of everything else, from 0x80 up (ASCII bytes are told apart directly).
// next[k][byte]: >= 1 the walk's next state, kWord / kOther: the character is complete; state 0 = a character begins.
struct WordClassifier {
  enum { kWord = -1, kOther = -2 };
  std::vector<std::array<int, 256>> next;
  std::vector<std::pair<uint8_t, uint8_t>> byte_ranges;  // what the walk tells apart (for the byte classes)
  bool build() {
    ClassSet word;
    if (!table_lookup("perl_word", &word)) return false;
    ClassSet w, o, not_word = negate(word);
    for (const Range &r : word)
      if (r.hi >= 0x80) w.push_back({std::max<uint32_t>(r.lo, 0x80), r.hi});
    for (const Range &r : not_word)
      if (r.hi >= 0x80) o.push_back({std::max<uint32_t>(r.lo, 0x80), r.hi});
    Nfa cn;
    const int tw = cn.add(NState::kMatch), to = cn.add(NState::kMatch);
    auto head = [&](const ClassSet &cs, int term) -> int {
      if (cs.empty()) return -1;
      return cn.instantiate(cn.class_dag(cs), term);
    };
    const int hw = head(w, tw), ho = head(o, to);
    if (cn.overflow) return false;
    for (const NState &st : cn.st)
      if (st.type == NState::kByte && st.lo <= st.hi) byte_ranges.push_back({st.lo, st.hi});
    auto close = [&](std::vector<int> seeds) {
      std::vector<int> out, stack = std::move(seeds);
      std::vector<char> seen(cn.st.size(), 0);
      while (!stack.empty()) {
        const int s = stack.back();
        stack.pop_back();
        if (s < 0 || seen[s]) continue;
        seen[s] = 1;
        const NState &n = cn.st[s];
        if (n.type == NState::kSplit) {
          stack.push_back(n.out);
          stack.push_back(n.out1);
        } else {
          out.push_back(s);
        }
      }
      std::sort(out.begin(), out.end());
      return out;
    };
    std::map<std::vector<int>, int> ids;
    std::vector<std::vector<int>> sets;
    sets.push_back(close({hw, ho}));
    ids[sets[0]] = 0;
    for (size_t cur = 0; cur < sets.size(); cur++) {
      std::array<int, 256> row;
      for (int b = 0; b < 256; b++) {
        std::vector<int> seeds;
        for (int s : sets[cur]) {
          const NState &n = cn.st[s];
          if (n.type == NState::kByte && n.lo <= b && b <= n.hi) seeds.push_back(n.out);
        }
        const std::vector<int> nx = close(seeds);
        if (nx.empty()) {
          row[b] = kOther;  // (not UTF-8: an Arrow string never holds it; ends the character)
        } else if (std::find(nx.begin(), nx.end(), tw) != nx.end()) {
          row[b] = kWord;
        } else if (std::find(nx.begin(), nx.end(), to) != nx.end()) {
          row[b] = kOther;
        } else {
          auto it = ids.find(nx);
          if (it == ids.end()) {
            it = ids.emplace(nx, (int)sets.size()).first;
            sets.push_back(nx);
          }
          row[b] = it->second;
        }
      }
      next.push_back(row);
      if (sets.size() > 4096) return false;
    }
    return true;
  }
};

}  // namespace

CompileStatus validate_pattern_rules(const char *pattern, size_t len, std::string *msg) {
  // SqlSecurity::validate_regex_pattern, TG/security.rs:152-183 and check_redos_patterns :258-281
  if (len > 1000) {
    *msg = "Regex pattern too long (max 1000 characters)";
    return kRejected;
  }
  if (memchr(pattern, 0, len) != nullptr) {
    *msg = "Regex pattern cannot contain null bytes";
    return kRejected;
  }
  static const char *const dangerous[] = {"(.*)*", "(.*)+", "(a+)+", "(a*)*"};
  std::string p(pattern, len);
  for (const char *d : dangerous)
    if (p.find(d) != std::string::npos) {
      *msg = "Regex pattern might cause ReDoS attack";
      return kRejected;
    }
  return kOk;
}

CompileStatus compile(const char *pattern, size_t len, bool case_insensitive, Dfa *out, std::string *msg) {
  Parser ps;
  if (!Parser::decode_utf8(pattern, len, &ps.p)) {
    *msg = "Invalid regex pattern: pattern is not valid UTF-8";
    return kInvalid;
  }
  Flags f;
  f.i = case_insensitive;
  NodeP ast = ps.parse_alt(f);
  if (ps.status == kOk && !ps.eof()) {
    if (ps.peek() == ')')
      ps.fail(kInvalid, "unopened group");
    else
      ps.fail(kInvalid, "unexpected character");
  }
  if (ps.status != kOk) {
    *msg = (ps.status == kInvalid ? "Invalid regex pattern: " : "") + ps.msg;
    return ps.status;
  }
  Nfa nfa;
  out->len_min = out->len_max = -1;
  {
    // `^C{m,n}$` with a class whose copies would make a big automaton: `^C*$` and a character count (regex_compile.h)
    auto strip = [](Node *n) -> Node * {
      while ((n->kind == Node::kConcat || n->kind == Node::kAlt) && n->kids.size() == 1) n = n->kids[0].get();
      return n;
    };
    Node *top = strip(ast.get());
    if (top->kind == Node::kConcat) {
      std::vector<Node *> parts;
      for (auto &k : top->kids)
        if (strip(k.get())->kind != Node::kEmpty) parts.push_back(strip(k.get()));
      if (parts.size() == 3 && parts[0]->kind == Node::kStart && parts[2]->kind == Node::kEnd &&
          parts[1]->kind == Node::kRepeat && parts[1]->max >= 0 && strip(parts[1]->kids[0].get())->kind == Node::kClass) {
        Node *rep = parts[1];
        const Nfa::ClassDag &dag = nfa.class_dag(strip(rep->kids[0].get())->cls);
        if ((size_t)rep->max * std::max<size_t>(dag.nodes.size(), 1) > 1024) {
          out->len_min = rep->min;
          out->len_max = rep->max;
          rep->min = 0;
          rep->max = -1;
        }
      }
    }
  }
  int match = nfa.add(NState::kMatch);
  int start = nfa.compile(ast.get(), match);
  if (nfa.overflow) {
    *msg = "pattern expands to too many NFA states for the GPU engine";
    return kUnsupported;
  }

  // which neighbourhood the pattern's assertions look at
  bool uses_line = false, uses_word = false, uses_uword = false;
  for (const NState &s : nfa.st) {
    uses_line |= s.type == NState::kAssertLineStart || s.type == NState::kAssertLineEnd;
    uses_word |= s.type == NState::kAssertWordB || s.type == NState::kAssertNotWordB;
    uses_uword |= s.type == NState::kAssertUWordB || s.type == NState::kAssertUNotWordB;
  }
  if (uses_word && uses_uword) {
    *msg = "ASCII and Unicode word boundaries in one pattern are not supported by the GPU engine";
    return kUnsupported;
  }
  // (the classifier does not depend on the pattern: built once per process)
  static WordClassifier wc_shared;
  static bool wc_ok = false;
  static std::once_flag wc_once;
  if (uses_uword) std::call_once(wc_once, [] { wc_ok = wc_shared.build(); });
  if (uses_uword && !wc_ok) {
    *msg = "Unicode word boundaries: the classifier of \\w could not be built";
    return kUnsupported;
  }
  const WordClassifier &wc = wc_shared;
  // byte classes from every byte-range boundary (and from the bytes the assertions tell apart)
  bool boundary[257];
  memset(boundary, 0, sizeof(boundary));
  boundary[0] = true;
  for (const NState &s : nfa.st)
    if (s.type == NState::kByte && s.lo <= s.hi) {
      boundary[s.lo] = true;
      boundary[(int)s.hi + 1] = true;
    }
  if (uses_line) boundary['\n'] = boundary['\n' + 1] = true;
  if (uses_word || uses_uword)
    for (int b = 1; b < 256; b++)
      if (Builder::is_word_byte(b) != Builder::is_word_byte(b - 1)) boundary[b] = true;
  if (uses_uword) {
    boundary[0x80] = true;
    for (const auto &r : wc.byte_ranges) {
      boundary[r.first] = true;
      boundary[(int)r.second + 1] = true;
    }
  }
  int ncls = 0;
  uint8_t rep[256];
  for (int b = 0; b < 256; b++) {
    if (boundary[b]) {
      rep[ncls] = (uint8_t)b;
      ncls++;
    }
    out->byte_class[b] = (uint8_t)(ncls - 1);
  }

  Builder bld(nfa);
  // the context a consumed byte leaves behind, no finer than the assertions need (a pattern without look-behind
  // assertions keeps ONE context, and its automaton is the one it always was)
  auto ctx_of = [&](int byte) -> int {
    if (uses_line && byte == '\n') return Builder::kCtxNewline;
    if ((uses_word || uses_uword) && Builder::is_word_byte(byte)) return Builder::kCtxWord;
    return Builder::kCtxOther;
  };
  // ... and, with Unicode word boundaries, the walk through a character's bytes: contexts 0..3 are character
  // boundaries, 4 + k is state k of the classifier (inside a character)
  auto ctx_step = [&](int ctx, int byte) -> int {
    if (!uses_uword) return ctx_of(byte);
    const int k = ctx >= 4 ? ctx - 4 : 0;
    if (k == 0 && byte < 0x80) return ctx_of(byte);
    const int r = wc.next[(size_t)k][(size_t)byte];
    if (r == WordClassifier::kWord) return Builder::kCtxWord;
    if (r == WordClassifier::kOther) return Builder::kCtxOther;
    return 4 + r;
  };
  // an unanchored search may begin a match at every position: the start state's closure under each context
  std::vector<int> init_unanchored[4];
  for (int c = 0; c < 4; c++) bld.closure({start}, c, Builder::kNextUnknown, &init_unanchored[c]);

  bool can_restart = false;
  for (int c = 1; c < 4; c++) can_restart |= !init_unanchored[c].empty();
  // a DFA state = (context of the byte before, set of NFA states); the context rides in front of the set as -(10 + c)
  std::map<std::vector<int>, int> ids;
  std::vector<std::vector<int>> sets;
  // ids: 0 = DEAD (empty set), 1 = MATCHED
  sets.push_back({});
  ids[{}] = 0;
  sets.push_back({-1});
  auto intern = [&](int ctx, const std::vector<int> &set) -> int {
    if (bld.has_match(set)) return 1;
    // an empty set is DEAD only if no later context can start a match (after a line feed (?m)^ can): states that
    // never reach an accept are folded into DEAD by the reachability pass below either way
    if (set.empty() && !can_restart) return 0;
    std::vector<int> key;
    key.reserve(set.size() + 1);
    key.push_back(-(10 + ctx));
    key.insert(key.end(), set.begin(), set.end());
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    int id = (int)sets.size();
    ids[key] = id;
    sets.push_back(key);
    return id;
  };
  int start_id = intern(Builder::kCtxStart, init_unanchored[Builder::kCtxStart]);
  // (see the loop below)
  bool plain_sets = true;
  // (`\A` is decided from the context inside the closure that follows a byte; a pending `$` / `\z` simply does not
  //  survive a byte: both leave the walk below as it is.  Line and word assertions take the general path.)
  for (const NState &ns : nfa.st)
    plain_sets &= ns.type == NState::kByte || ns.type == NState::kSplit || ns.type == NState::kEmpty ||
                  ns.type == NState::kMatch || ns.type == NState::kAssertStart || ns.type == NState::kAssertEnd;
  std::vector<std::vector<int>> seeds_by_class((size_t)ncls);
  std::map<std::vector<int>, int> closed_before;
  std::vector<int> memo_key;
  std::vector<uint16_t> table;
  std::vector<uint8_t> acc_end;
  for (size_t cur = 0; cur < sets.size(); cur++) {
    if (sets.size() > kMaxDfaStates) {
      *msg = "pattern needs more than 60000 DFA states; not supported by the GPU engine";
      return kUnsupported;
    }
    table.resize((cur + 1) * (size_t)ncls);
    acc_end.resize(cur + 1);
    if (cur == 0) {
      for (int c = 0; c < ncls; c++) table[c] = 0;
      acc_end[0] = 0;
      continue;
    }
    if (cur == 1) {
      for (int c = 0; c < ncls; c++) table[ncls + c] = 1;
      acc_end[1] = 1;
      continue;
    }
    const int ctx = -sets[cur][0] - 10;
    const std::vector<int> set(sets[cur].begin() + 1, sets[cur].end());
    {
      // the haystack ends here: the pending look-ahead assertions are decided with "no next byte"
      std::vector<int> endc;
      bld.closure(set, ctx, Builder::kNextEnd, &endc);
      acc_end[cur] = bld.has_match(endc) ? 1 : 0;
    }
    if (plain_sets) {
      // No assertion anywhere in the pattern (the common case, and every big automaton there is): the set IS its own
      // closure whatever byte follows, so the byte states are handed to the classes their ranges cover in ONE walk over
      // the set, and a (context, seeds) pair that has been closed before -- most classes of most states carry the same
      // few seeds, or none -- is looked up instead of closed again.  (`\w{3,40}@\w+`: 14.2 s -> well under a second.)
      for (int c = 0; c < ncls; c++) seeds_by_class[(size_t)c].clear();
      for (int e : set) {
        const NState &n = nfa.st[bld.state_of(e)];
        if (n.type != NState::kByte || n.lo > n.hi || n.out < 0) continue;
        const int c0 = out->byte_class[n.lo], c1 = out->byte_class[n.hi];
        for (int c = c0; c <= c1; c++) seeds_by_class[(size_t)c].push_back(n.out);
      }
      table.resize(std::max(table.size(), (cur + 1) * (size_t)ncls));
      for (int c = 0; c < ncls; c++) {
        const int nctx = ctx_step(ctx, rep[c]);
        std::vector<int> &seeds = seeds_by_class[(size_t)c];
        std::sort(seeds.begin(), seeds.end());
        seeds.erase(std::unique(seeds.begin(), seeds.end()), seeds.end());
        memo_key.clear();
        memo_key.push_back(nctx);
        memo_key.insert(memo_key.end(), seeds.begin(), seeds.end());
        auto hit = closed_before.find(memo_key);
        int id;
        if (hit != closed_before.end()) {
          id = hit->second;
        } else {
          std::vector<int> next;
          bld.closure(seeds, nctx, Builder::kNextUnknown, &next);
          next.insert(next.end(), init_unanchored[nctx].begin(), init_unanchored[nctx].end());
          std::sort(next.begin(), next.end());
          next.erase(std::unique(next.begin(), next.end()), next.end());
          id = intern(nctx, next);
          closed_before.emplace(memo_key, id);
        }
        table[cur * ncls + c] = (uint16_t)id;
      }
      continue;
    }
    for (int c = 0; c < ncls; c++) {
      const uint8_t byte = rep[c];
      // 1. the pending look-ahead assertions of this position, now that the next byte is known
      std::vector<int> here;
      bld.closure(set, ctx, (int)byte, &here);
      int id;
      if (bld.has_match(here)) {
        id = 1;  // matched before this byte
      } else {
        // 2. the byte itself, then the closure behind it (context = this byte), plus a fresh start
        std::vector<int> seeds;
        for (int e : here) {
          const NState &n = nfa.st[bld.state_of(e)];
          const int ob = bld.obligation_of(e);
          if (n.type == NState::kByte && n.lo <= byte && byte <= n.hi && n.out >= 0) seeds.push_back(bld.enc(n.out, ob));
          // (a match that still waits for the character behind it to turn out right stays until that is known)
          if (n.type == NState::kMatch && ob != Builder::kNoObligation) seeds.push_back(e);
        }
        const int nctx = ctx_step(ctx, byte);
        if (uses_uword && nctx < 4) {  // a character is complete: what the threads required of it is settled
          const bool was_word = nctx == Builder::kCtxWord;
          std::vector<int> kept;
          for (int e : seeds) {
            const int ob = bld.obligation_of(e);
            if (ob == Builder::kNoObligation)
              kept.push_back(e);
            else if ((ob == Builder::kNeedWord) == was_word)
              kept.push_back(bld.enc(bld.state_of(e), Builder::kNoObligation));
          }
          seeds.swap(kept);
        }
        std::vector<int> next;
        bld.closure(seeds, nctx, Builder::kNextUnknown, &next);
        // (a match may begin at every CHARACTER: inside one -- contexts from 4 up -- none does)
        if (nctx < 4) next.insert(next.end(), init_unanchored[nctx].begin(), init_unanchored[nctx].end());
        std::sort(next.begin(), next.end());
        next.erase(std::unique(next.begin(), next.end()), next.end());
        id = intern(nctx, next);
      }
      table.resize(std::max(table.size(), (cur + 1) * (size_t)ncls));
      table[cur * ncls + c] = (uint16_t)id;
    }
  }
  const int n_states = (int)sets.size();
  if (getenv("TGX_RX_TIMING")) fprintf(stderr, "rx: nfa %zu states, subset construction -> %d states x %d classes (clock %.2f s)\n", nfa.st.size(), n_states, ncls, (double)clock() / CLOCKS_PER_SEC);
  // states from which neither MATCHED nor an accept-at-end state is reachable are DEAD
  std::vector<char> alive(n_states, 0);
  std::vector<std::vector<int>> rev(n_states);
  for (int s = 0; s < n_states; s++)
    for (int c = 0; c < ncls; c++) rev[table[(size_t)s * ncls + c]].push_back(s);
  std::vector<int> work;
  for (int s = 0; s < n_states; s++)
    if (s == 1 || acc_end[s]) {
      alive[s] = 1;
      work.push_back(s);
    }
  while (!work.empty()) {
    int s = work.back();
    work.pop_back();
    for (int p : rev[s])
      if (!alive[p]) {
        alive[p] = 1;
        work.push_back(p);
      }
  }
  for (auto &t : table)
    if (!alive[t]) t = 0;
  // ---- minimise: Moore partition refinement (DEAD and MATCHED keep ids 0 and 1), then merge byte classes
  // whose columns are identical.  Unicode classes expand to many byte-sequence states that are
  // equivalent (`^\\d{5}$`: 403 -> a few dozen states), which is what makes most tables fit in LDS.
  std::vector<int> part(n_states);
  for (int s = 0; s < n_states; s++) part[s] = s == 0 ? 0 : s == 1 ? 1 : (acc_end[s] ? 2 : 3);
  int n_parts = 0;  // partitions after the previous round (the initial labelling may leave ids unused)
  for (;;) {
    std::map<std::vector<int>, int> sig_ids;
    std::vector<int> next(n_states);
    // keep 0 and 1 fixed
    sig_ids[{-1}] = 0;
    sig_ids[{-2}] = 1;
    std::vector<int> sig((size_t)ncls + 1);
    for (int s = 0; s < n_states; s++) {
      if (s == 0) { next[s] = 0; continue; }
      if (s == 1) { next[s] = 1; continue; }
      sig[0] = part[s];
      for (int c = 0; c < ncls; c++) sig[(size_t)c + 1] = part[table[(size_t)s * ncls + c]];
      auto it = sig_ids.find(sig);
      if (it == sig_ids.end()) it = sig_ids.emplace(sig, (int)sig_ids.size()).first;
      next[s] = it->second;
    }
    const int n_next = (int)sig_ids.size();
    part.swap(next);
    if (n_next == n_parts) break;
    n_parts = n_next;
  }
  // a partition id may be unused (e.g. no accept-at-end states): compact ids, keeping 0 and 1
  std::vector<int> remap(n_parts, -1);
  remap[0] = 0;
  remap[1] = 1;
  int m_states = 2;
  for (int s = 0; s < n_states; s++)
    if (remap[part[s]] < 0) remap[part[s]] = m_states++;
  std::vector<uint16_t> mtable((size_t)m_states * ncls, 0);
  std::vector<uint8_t> macc(m_states, 0);
  for (int c = 0; c < ncls; c++) mtable[(size_t)ncls + c] = 1;
  macc[1] = 1;
  for (int s = 2; s < n_states; s++) {
    const int q = remap[part[s]];
    macc[q] = acc_end[s];
    for (int c = 0; c < ncls; c++) mtable[(size_t)q * ncls + c] = (uint16_t)remap[part[table[(size_t)s * ncls + c]]];
  }
  // merge identical columns
  std::map<std::vector<uint16_t>, int> col_ids;
  std::vector<int> cls_map(ncls);
  for (int c = 0; c < ncls; c++) {
    std::vector<uint16_t> col(m_states);
    for (int q = 0; q < m_states; q++) col[q] = mtable[(size_t)q * ncls + c];
    auto it = col_ids.find(col);
    if (it == col_ids.end()) it = col_ids.emplace(col, (int)col_ids.size()).first;
    cls_map[c] = it->second;
  }
  const int m_cls = (int)col_ids.size();
  std::vector<uint16_t> ftable((size_t)m_states * m_cls);
  for (int q = 0; q < m_states; q++)
    for (int c = 0; c < ncls; c++) ftable[(size_t)q * m_cls + cls_map[c]] = mtable[(size_t)q * ncls + c];
  for (int b = 0; b < 256; b++) out->byte_class[b] = (uint8_t)cls_map[out->byte_class[b]];
  if (getenv("TGX_RX_TIMING")) fprintf(stderr, "rx: minimised -> %d states x %d classes (clock %.2f s)\n", m_states, m_cls, (double)clock() / CLOCKS_PER_SEC);
  out->n_states = (uint32_t)m_states;
  out->n_classes = (uint32_t)m_cls;
  out->start = alive[start_id] ? (uint32_t)remap[part[start_id]] : 0u;
  out->table = ftable;
  out->accept_at_end = macc;
  return kOk;
}

bool dfa_is_match(const Dfa &d, const uint8_t *s, size_t n) {
  if (d.len_max >= 0) {  // characters = bytes that are not continuation bytes
    int64_t chars = 0;
    for (size_t i = 0; i < n; i++) chars += (s[i] & 0xC0) != 0x80;
    if (chars < d.len_min || chars > d.len_max) return false;
  }
  uint32_t st = d.start;
  for (size_t i = 0; i < n && st > 1; i++) st = d.table[(size_t)st * d.n_classes + d.byte_class[s[i]]];
  return st == 1 || d.accept_at_end[st] != 0;
}

bool dfa_product(const std::vector<const Dfa *> &parts, uint32_t max_entries, ProductDfa *out) {
  const size_t k = parts.size();
  if (k == 0 || k > 8) return false;
  for (const Dfa *p : parts)
    if (p->len_max >= 0) return false;  // (an automaton with a character count is walked on its own)
  // common refinement of the byte classes: bytes with the same class in every part
  std::map<std::vector<uint8_t>, uint32_t> class_ids;
  std::vector<std::vector<uint8_t>> class_tuple;
  for (uint32_t b = 0; b < 256; b++) {
    std::vector<uint8_t> t(k);
    for (size_t i = 0; i < k; i++) t[i] = parts[i]->byte_class[b];
    auto it = class_ids.find(t);
    if (it == class_ids.end()) {
      it = class_ids.emplace(t, (uint32_t)class_tuple.size()).first;
      class_tuple.push_back(t);
    }
    out->byte_class[b] = (uint8_t)it->second;
    if (class_tuple.size() > 255) return false;
  }
  const uint32_t ncls = (uint32_t)class_tuple.size();
  const uint32_t n_final = 1u << k;
  auto decided = [](uint16_t s) { return s <= 1; };
  std::map<std::vector<uint16_t>, uint32_t> ids;
  std::vector<std::vector<uint16_t>> tuples;
  // the 2^k decided tuples first: index = mask of the parts that matched
  for (uint32_t m = 0; m < n_final; m++) {
    std::vector<uint16_t> t(k);
    for (size_t i = 0; i < k; i++) t[i] = (m >> i) & 1u;
    ids[t] = m;
    tuples.push_back(t);
  }
  auto id_of = [&](const std::vector<uint16_t> &t) -> uint32_t {
    auto it = ids.find(t);
    if (it != ids.end()) return it->second;
    const uint32_t id = (uint32_t)tuples.size();
    ids[t] = id;
    tuples.push_back(t);
    return id;
  };
  std::vector<uint16_t> start(k);
  for (size_t i = 0; i < k; i++) start[i] = (uint16_t)parts[i]->start;
  out->start = id_of(start);
  std::vector<uint16_t> table;
  for (uint32_t sidx = 0; sidx < tuples.size(); sidx++) {
    if ((uint64_t)tuples.size() * ncls > max_entries || tuples.size() > 65000) return false;
    const std::vector<uint16_t> cur = tuples[sidx];  // (copy: id_of grows `tuples`)
    table.resize((size_t)(sidx + 1) * ncls);
    for (uint32_t c = 0; c < ncls; c++) {
      std::vector<uint16_t> nxt(k);
      for (size_t i = 0; i < k; i++) {
        const Dfa &d = *parts[i];
        nxt[i] = decided(cur[i]) ? cur[i] : d.table[(size_t)cur[i] * d.n_classes + class_tuple[c][i]];
      }
      table[(size_t)sidx * ncls + c] = (uint16_t)id_of(nxt);
    }
  }
  if ((uint64_t)tuples.size() * ncls > max_entries) return false;
  out->n_parts = (uint32_t)k;
  out->n_states = (uint32_t)tuples.size();
  out->n_classes = ncls;
  out->n_final = n_final;
  out->table = std::move(table);
  out->accept_mask.assign(tuples.size(), 0);
  for (size_t sidx = 0; sidx < tuples.size(); sidx++) {
    uint8_t m = 0;
    for (size_t i = 0; i < k; i++) {
      const uint16_t st = tuples[sidx][i];
      if (st == 1 || parts[i]->accept_at_end[st] != 0) m |= (uint8_t)(1u << i);
    }
    out->accept_mask[sidx] = m;
  }
  return true;
}

uint32_t product_match_mask(const ProductDfa &d, const uint8_t *s, size_t n) {
  uint32_t st = d.start;
  for (size_t i = 0; i < n && st >= d.n_final; i++) st = d.table[(size_t)st * d.n_classes + d.byte_class[s[i]]];
  return d.accept_mask[st];
}

}  // namespace rx
}  // namespace tgx
