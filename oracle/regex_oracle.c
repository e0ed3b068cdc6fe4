/*
 * regex_oracle.c -- CPU ORACLE for the pattern checks. TEST INFRASTRUCTURE ONLY (see tgx_oracle.h).
 *
 * The reference evaluates `col ~ 'pat'` through DataFusion -> arrow-string regexp_is_match ->
 * regex::Regex::is_match (regex 1.12.2 / regex-automata 0.4.13 / regex-syntax 0.8.8, Cargo.lock:3637-3661;
 * none of them is under /root/reference).  This file restates the published algorithm those crates
 * implement for is_match -- Thompson NFA + Pike-style lock-step simulation over Unicode scalar values,
 * leftmost-unanchored search -- with the syntax subset the reference's patterns use
 * (TG/constraints/format.rs:237-294) plus what regex-syntax documents around it.  It shares no code with
 * the product's DFA compiler (term_amd/csrc/regex): code points instead of bytes, simulation instead of
 * determinisation, its own parser.
 *
 * Pinned by: the reference's format vectors (tests/golden/reference_vectors.json) and an independent
 * cross-check against the `regex` PyPI module (tests/golden/regex_crosscheck.json).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tgx_oracle.h"
#include "unicode_oracle_tables.h"

#define MAXCP 0x10FFFFu

typedef struct { uint32_t lo, hi; } rng_t;
typedef struct { rng_t *r; int n, cap; } cls_t;

static void cls_add(cls_t *c, uint32_t lo, uint32_t hi) {
  if (c->n == c->cap) {
    c->cap = c->cap ? c->cap * 2 : 8;
    c->r = (rng_t *)realloc(c->r, (size_t)c->cap * sizeof(rng_t));
  }
  c->r[c->n].lo = lo;
  c->r[c->n].hi = hi;
  c->n++;
}
static int rng_cmp(const void *a, const void *b) {
  uint32_t x = ((const rng_t *)a)->lo, y = ((const rng_t *)b)->lo;
  return x < y ? -1 : x > y;
}
static void cls_norm(cls_t *c) {
  if (c->n == 0) return;
  qsort(c->r, (size_t)c->n, sizeof(rng_t), rng_cmp);
  int w = 0;
  for (int i = 1; i < c->n; i++) {
    if (c->r[i].lo <= c->r[w].hi + 1 && c->r[w].hi != MAXCP + 1) {
      if (c->r[i].hi > c->r[w].hi) c->r[w].hi = c->r[i].hi;
    } else {
      c->r[++w] = c->r[i];
    }
  }
  c->n = w + 1;
}
static void cls_negate(cls_t *c) {
  cls_norm(c);
  cls_t o = {0, 0, 0};
  uint32_t next = 0;
  for (int i = 0; i < c->n; i++) {
    if (c->r[i].lo > next) cls_add(&o, next, c->r[i].lo - 1);
    next = c->r[i].hi + 1;
  }
  if (next <= MAXCP) cls_add(&o, next, MAXCP);
  free(c->r);
  *c = o;
}
static int cls_has(const cls_t *c, uint32_t cp) {
  for (int i = 0; i < c->n; i++)
    if (c->r[i].lo <= cp && cp <= c->r[i].hi) return 1;
  return 0;
}
static void cls_union(cls_t *dst, const cls_t *src) {
  for (int i = 0; i < src->n; i++) cls_add(dst, src->r[i].lo, src->r[i].hi);
}
/* simple case folding: add every code point equivalent to a member */
static void cls_fold(cls_t *c) {
  cls_norm(c);
  int n0 = c->n;
  cls_t extra = {0, 0, 0};
  for (uint32_t i = 0; i < tgx_n_fold_pairs; i++) {
    uint32_t a = tgx_fold_pairs[i][0];
    for (int k = 0; k < n0; k++)
      if (c->r[k].lo <= a && a <= c->r[k].hi) {
        cls_add(&extra, tgx_fold_pairs[i][1], tgx_fold_pairs[i][1]);
        break;
      }
  }
  cls_union(c, &extra);
  free(extra.r);
  cls_norm(c);
}
static int cls_table(cls_t *c, const char *name) {
  for (uint32_t i = 0; i < tgx_n_utables; i++)
    if (strcmp(tgx_utables[i].name, name) == 0) {
      for (uint32_t k = 0; k < tgx_utables[i].count; k++)
        cls_add(c, tgx_utables[i].ranges[k].lo, tgx_utables[i].ranges[k].hi);
      return 1;
    }
  return 0;
}

/* ---------------------------------------------------------------- AST */
enum { N_EMPTY, N_CLASS, N_START, N_END, N_BOL, N_EOL, N_WORDB, N_NWORDB, N_UWORDB, N_UNWORDB, N_CAT, N_ALT, N_REP };
typedef struct node {
  int kind;
  cls_t cls;
  struct node **kid;
  int nkid, capkid;
  int min, max; /* N_REP, max < 0 = unbounded */
} node_t;

static node_t *node_new(int kind) {
  node_t *n = (node_t *)calloc(1, sizeof(node_t));
  n->kind = kind;
  return n;
}
static void node_push(node_t *p, node_t *k) {
  if (p->nkid == p->capkid) {
    p->capkid = p->capkid ? p->capkid * 2 : 4;
    p->kid = (node_t **)realloc(p->kid, (size_t)p->capkid * sizeof(node_t *));
  }
  p->kid[p->nkid++] = k;
}
static void node_free(node_t *n) {
  if (!n) return;
  for (int i = 0; i < n->nkid; i++) node_free(n->kid[i]);
  free(n->kid);
  free(n->cls.r);
  free(n);
}

typedef struct {
  int i, s, x;
  int m;      /* (?m): ^ / $ also match after / before a line feed */
  int ascii;  /* (?-u): \w \d \s \b \B are the ASCII ones */
} flags_t;

typedef struct {
  uint32_t *p;
  int n, pos;
  int failed;
  char msg[200];
  int depth;
  int ascii; /* the (?-u) flag of the group being parsed (class_escape has no flags argument) */
} parser_t;

static int p_fail(parser_t *ps, const char *m) {
  if (!ps->failed) {
    ps->failed = 1;
    snprintf(ps->msg, sizeof(ps->msg), "%s", m);
  }
  return 0;
}
static uint32_t p_peek(parser_t *ps, int k) { return ps->pos + k < ps->n ? ps->p[ps->pos + k] : 0xFFFFFFFFu; }
static int p_eof(parser_t *ps) { return ps->pos >= ps->n; }
static int p_eat(parser_t *ps, uint32_t c) {
  if (p_peek(ps, 0) == c) {
    ps->pos++;
    return 1;
  }
  return 0;
}

static node_t *parse_alt(parser_t *ps, flags_t *f);

static node_t *mk_literal(uint32_t cp, const flags_t *f) {
  node_t *n = node_new(N_CLASS);
  cls_add(&n->cls, cp, cp);
  if (f->i) cls_fold(&n->cls);
  return n;
}

static int hexv(uint32_t c) {
  if (c >= '0' && c <= '9') return (int)(c - '0');
  if (c >= 'a' && c <= 'f') return (int)(c - 'a') + 10;
  if (c >= 'A' && c <= 'F') return (int)(c - 'A') + 10;
  return -1;
}
static int parse_hex(parser_t *ps, int digits, uint32_t *out) {
  uint32_t v = 0;
  if (p_peek(ps, 0) == '{') {
    ps->pos++;
    int n = 0;
    while (p_peek(ps, 0) != '}') {
      int h = hexv(p_peek(ps, 0));
      if (h < 0 || n >= 8) return p_fail(ps, "invalid hexadecimal escape");
      v = v * 16 + (uint32_t)h;
      n++;
      ps->pos++;
    }
    ps->pos++;
    if (!n) return p_fail(ps, "empty hexadecimal escape");
  } else {
    for (int k = 0; k < digits; k++) {
      int h = hexv(p_peek(ps, 0));
      if (h < 0) return p_fail(ps, "invalid hexadecimal escape");
      v = v * 16 + (uint32_t)h;
      ps->pos++;
    }
  }
  if (v > MAXCP || (v >= 0xD800 && v <= 0xDFFF)) return p_fail(ps, "escape is not a Unicode scalar value");
  *out = v;
  return 1;
}

static void lower_strip(const char *in, char *out, size_t cap) {
  size_t w = 0;
  for (; *in && w + 1 < cap; in++) {
    char c = *in;
    if (c == ' ' || c == '_' || c == '-') continue;
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    out[w++] = c;
  }
  out[w] = 0;
}

/* \p{name}: general categories, a few binary properties, a few scripts */
static int unicode_prop(parser_t *ps, const char *name, cls_t *out) {
  static const char *const long_names[][2] = {
      {"letter", "l"}, {"uppercaseletter", "lu"}, {"lowercaseletter", "ll"}, {"titlecaseletter", "lt"},
      {"modifierletter", "lm"}, {"otherletter", "lo"}, {"mark", "m"}, {"nonspacingmark", "mn"},
      {"spacingmark", "mc"}, {"enclosingmark", "me"}, {"number", "n"}, {"decimalnumber", "nd"},
      {"letternumber", "nl"}, {"othernumber", "no"}, {"punctuation", "p"}, {"connectorpunctuation", "pc"},
      {"dashpunctuation", "pd"}, {"openpunctuation", "ps"}, {"closepunctuation", "pe"},
      {"initialpunctuation", "pi"}, {"finalpunctuation", "pf"}, {"otherpunctuation", "po"}, {"symbol", "s"},
      {"mathsymbol", "sm"}, {"currencysymbol", "sc"}, {"modifiersymbol", "sk"}, {"othersymbol", "so"},
      {"separator", "z"}, {"spaceseparator", "zs"}, {"lineseparator", "zl"}, {"paragraphseparator", "zp"},
      {"other", "c"}, {"control", "cc"}, {"format", "cf"}, {"privateuse", "co"}, {"unassigned", "cn"},
      {"digit", "nd"}, {"punct", "p"}};
  char key[64], lhs[64] = "";
  lower_strip(name, key, sizeof(key));
  char *eq = strchr(key, '=');
  if (eq) {
    *eq = 0;
    snprintf(lhs, sizeof(lhs), "%s", key);
    memmove(key, eq + 1, strlen(eq + 1) + 1);
  }
  for (size_t i = 0; i < sizeof(long_names) / sizeof(long_names[0]); i++)
    if (strcmp(key, long_names[i][0]) == 0) snprintf(key, sizeof(key), "%s", long_names[i][1]);
  if (strcmp(key, "any") == 0) {
    cls_add(out, 0, MAXCP);
    return 1;
  }
  if (strcmp(key, "ascii") == 0) {
    cls_add(out, 0, 0x7F);
    return 1;
  }
  for (uint32_t i = 0; i < tgx_n_utables; i++) {
    const char *t = tgx_utables[i].name;
    const char *us = strchr(t, '_');
    if (!us || strncmp(t, "perl_", 5) == 0) continue;
    char val[64];
    lower_strip(us + 1, val, sizeof(val));
    if (strcmp(val, key) != 0) continue;
    if (lhs[0]) {
      int is_gc = strncmp(t, "gc_", 3) == 0, is_sc = strncmp(t, "script_", 7) == 0;
      int ok = (is_gc && (!strcmp(lhs, "gc") || !strcmp(lhs, "generalcategory"))) ||
               (is_sc && (!strcmp(lhs, "sc") || !strcmp(lhs, "script")));
      if (!ok) continue;
    }
    for (uint32_t k = 0; k < tgx_utables[i].count; k++)
      cls_add(out, tgx_utables[i].ranges[k].lo, tgx_utables[i].ranges[k].hi);
    return 1;
  }
  return p_fail(ps, "unsupported Unicode property");
}

/* handles \d \D \s \S \w \W \p \P after the backslash char `c` has been consumed; 1 = was a class */
static int class_escape(parser_t *ps, uint32_t c, cls_t *out, int *is_class) {
  *is_class = 1;
  if (ps->ascii) { /* (?-u): ASCII Perl classes; the negated ones could match invalid UTF-8 (regex-syntax rejects them) */
    switch (c) {
      case 'd': cls_add(out, '0', '9'); return 1;
      case 's': cls_add(out, '\t', '\r'); cls_add(out, ' ', ' '); cls_norm(out); return 1;
      case 'w': cls_add(out, '0', '9'); cls_add(out, 'A', 'Z'); cls_add(out, '_', '_'); cls_add(out, 'a', 'z'); cls_norm(out); return 1;
      case 'D': case 'S': case 'W': case 'p': case 'P': return p_fail(ps, "negated / Unicode classes under (?-u)");
      default: break;
    }
  }
  switch (c) {
    case 'd': cls_table(out, "perl_digit"); return 1;
    case 'D': cls_table(out, "perl_digit"); cls_negate(out); return 1;
    case 's': cls_table(out, "perl_space"); return 1;
    case 'S': cls_table(out, "perl_space"); cls_negate(out); return 1;
    case 'w': cls_table(out, "perl_word"); return 1;
    case 'W': cls_table(out, "perl_word"); cls_negate(out); return 1;
    case 'p':
    case 'P': {
      char name[64];
      size_t w = 0;
      int neg = c == 'P';
      if (p_peek(ps, 0) == '{') {
        ps->pos++;
        if (p_peek(ps, 0) == '^') {
          neg = !neg;
          ps->pos++;
        }
        while (!p_eof(ps) && p_peek(ps, 0) != '}' && w + 1 < sizeof(name)) name[w++] = (char)ps->p[ps->pos++];
        if (!p_eat(ps, '}')) return p_fail(ps, "unclosed Unicode class");
      } else {
        if (p_eof(ps)) return p_fail(ps, "incomplete escape");
        name[w++] = (char)ps->p[ps->pos++];
      }
      name[w] = 0;
      if (!unicode_prop(ps, name, out)) return 0;
      if (neg) cls_negate(out);
      return 1;
    }
    default:
      *is_class = 0;
      return 1;
  }
}

static int simple_escape(parser_t *ps, uint32_t c, uint32_t *cp, int *handled) {
  *handled = 1;
  switch (c) {
    case 'n': *cp = '\n'; return 1;
    case 't': *cp = '\t'; return 1;
    case 'r': *cp = '\r'; return 1;
    case 'f': *cp = '\f'; return 1;
    case 'v': *cp = 0x0B; return 1;
    case 'a': *cp = 0x07; return 1;
    case 'x': return parse_hex(ps, 2, cp);
    case 'u': return parse_hex(ps, 4, cp);
    case 'U': return parse_hex(ps, 8, cp);
    case '0': return p_fail(ps, "octal escapes are not supported");
    default: break;
  }
  int alnum = (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
  if (c < 0x80 && c >= ' ' && !alnum) {
    *cp = c;
    return 1;
  }
  *handled = 0;
  return 1;
}

static int posix_class(const char *name, cls_t *out) {
  struct { const char *n; rng_t r[4]; int k; } t[] = {
      {"alnum", {{'0', '9'}, {'A', 'Z'}, {'a', 'z'}}, 3}, {"alpha", {{'A', 'Z'}, {'a', 'z'}}, 2},
      {"ascii", {{0, 0x7F}}, 1}, {"blank", {{'\t', '\t'}, {' ', ' '}}, 2}, {"cntrl", {{0, 0x1F}, {0x7F, 0x7F}}, 2},
      {"digit", {{'0', '9'}}, 1}, {"graph", {{'!', '~'}}, 1}, {"lower", {{'a', 'z'}}, 1}, {"print", {{' ', '~'}}, 1},
      {"punct", {{'!', '/'}, {':', '@'}, {'[', '`'}, {'{', '~'}}, 4}, {"space", {{'\t', '\r'}, {' ', ' '}}, 2},
      {"upper", {{'A', 'Z'}}, 1}, {"word", {{'0', '9'}, {'A', 'Z'}, {'_', '_'}, {'a', 'z'}}, 4},
      {"xdigit", {{'0', '9'}, {'A', 'F'}, {'a', 'f'}}, 3}};
  for (size_t i = 0; i < sizeof(t) / sizeof(t[0]); i++)
    if (strcmp(name, t[i].n) == 0) {
      for (int k = 0; k < t[i].k; k++) cls_add(out, t[i].r[k].lo, t[i].r[k].hi);
      return 1;
    }
  return 0;
}

static int class_item(parser_t *ps, uint32_t *cp, cls_t *cs, int *is_class) {
  *is_class = 0;
  uint32_t c = ps->p[ps->pos++];
  if (c != '\\') {
    *cp = c;
    return 1;
  }
  if (p_eof(ps)) return p_fail(ps, "incomplete escape");
  uint32_t e = ps->p[ps->pos++];
  if (e == 'b') {
    *cp = 0x08;
    return 1;
  }
  if (!class_escape(ps, e, cs, is_class)) return 0;
  if (*is_class) return 1;
  int handled = 0;
  if (!simple_escape(ps, e, cp, &handled)) return 0;
  if (!handled) return p_fail(ps, "unrecognized escape sequence");
  return 1;
}

/* a && b as the complement of (not a) union (not b) */
static void cls_intersect(cls_t *a, const cls_t *b) {
  cls_t nb = {0, 0, 0};
  cls_union(&nb, b);
  cls_norm(&nb);
  cls_negate(&nb);
  cls_negate(a);
  cls_union(a, &nb);
  cls_norm(a);
  cls_negate(a);
  free(nb.r);
}
/* regex-syntax's class set operations: && -- ~~ bind weaker than the union of adjacent items, left to right */
static void cls_apply(cls_t *result, int op, cls_t *acc, const flags_t *f) {
  cls_norm(acc);
  if (f->i) cls_fold(acc);
  if (op == 0) {
    free(result->r);
    *result = *acc;
    acc->r = NULL;
    acc->n = acc->cap = 0;
    return;
  }
  if (op == '&') {
    cls_intersect(result, acc);
  } else if (op == '-') {
    cls_t nb = {0, 0, 0};
    cls_union(&nb, acc);
    cls_norm(&nb);
    cls_negate(&nb);
    cls_intersect(result, &nb);
    free(nb.r);
  } else { /* ~~ : (a or b) minus (a and b) */
    cls_t both = {0, 0, 0}, any = {0, 0, 0};
    cls_union(&both, result);
    cls_norm(&both);
    cls_intersect(&both, acc);
    cls_union(&any, result);
    cls_union(&any, acc);
    cls_norm(&any);
    cls_negate(&both);
    cls_intersect(&any, &both);
    free(result->r);
    *result = any;
    free(both.r);
  }
  free(acc->r);
  acc->r = NULL;
  acc->n = acc->cap = 0;
}

static int parse_class(parser_t *ps, const flags_t *f, cls_t *out) {
  ps->ascii = f->ascii;
  ps->pos++; /* [ */
  int neg = p_eat(ps, '^'), first = 1, pending_op = 0;
  cls_t acc = {0, 0, 0}, result = {0, 0, 0};
  for (;;) {
    if (p_eof(ps)) {
      free(acc.r);
      return p_fail(ps, "unclosed character class");
    }
    uint32_t c = p_peek(ps, 0);
    if (c == ']' && !first) {
      ps->pos++;
      break;
    }
    first = 0;
    if (c == '[') {
      if (p_peek(ps, 1) == ':') {
        int save = ps->pos;
        ps->pos += 2;
        int pneg = p_eat(ps, '^');
        char name[20];
        size_t w = 0;
        while (!p_eof(ps) && p_peek(ps, 0) != ':' && w + 1 < sizeof(name)) name[w++] = (char)ps->p[ps->pos++];
        name[w] = 0;
        cls_t pc = {0, 0, 0};
        if (p_peek(ps, 0) == ':' && p_peek(ps, 1) == ']' && posix_class(name, &pc)) {
          ps->pos += 2;
          if (pneg) cls_negate(&pc);
          cls_union(&acc, &pc);
          free(pc.r);
          continue;
        }
        free(pc.r);
        ps->pos = save;
      }
      cls_t nested = {0, 0, 0};
      if (!parse_class(ps, f, &nested)) {
        free(acc.r);
        return 0;
      }
      cls_union(&acc, &nested);
      free(nested.r);
      continue;
    }
    if ((c == '&' && p_peek(ps, 1) == '&') || (c == '-' && p_peek(ps, 1) == '-') ||
        (c == '~' && p_peek(ps, 1) == '~')) {
      cls_apply(&result, pending_op, &acc, f);
      pending_op = (int)c;
      ps->pos += 2;
      continue;
    }
    uint32_t lo = 0;
    int is_class = 0;
    cls_t cs = {0, 0, 0};
    if (!class_item(ps, &lo, &cs, &is_class)) {
      free(acc.r);
      free(cs.r);
      return 0;
    }
    if (is_class) {
      cls_union(&acc, &cs);
      free(cs.r);
      continue;
    }
    if (p_peek(ps, 0) == '-' && p_peek(ps, 1) != ']' && p_peek(ps, 1) != '-' && !p_eof(ps)) {
      ps->pos++;
      uint32_t hi = 0;
      int hic = 0;
      cls_t hs = {0, 0, 0};
      if (!class_item(ps, &hi, &hs, &hic)) {
        free(acc.r);
        free(hs.r);
        return 0;
      }
      free(hs.r);
      if (hic || hi < lo) {
        free(acc.r);
        return p_fail(ps, "invalid character class range");
      }
      cls_add(&acc, lo, hi);
    } else {
      cls_add(&acc, lo, lo);
    }
  }
  cls_apply(&result, pending_op, &acc, f);
  if (neg) cls_negate(&result);
  *out = result;
  return 1;
}

static void skip_ws(parser_t *ps, const flags_t *f) {
  if (!f->x) return;
  for (;;) {
    uint32_t c = p_peek(ps, 0);
    if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == 0x0B)
      ps->pos++;
    else if (c == '#')
      while (!p_eof(ps) && p_peek(ps, 0) != '\n') ps->pos++;
    else
      break;
  }
}

static int parse_dec(parser_t *ps, int *out) {
  if (!(p_peek(ps, 0) >= '0' && p_peek(ps, 0) <= '9')) return 0;
  long v = 0;
  while (p_peek(ps, 0) >= '0' && p_peek(ps, 0) <= '9') {
    v = v * 10 + (long)(p_peek(ps, 0) - '0');
    if (v > 100000) return p_fail(ps, "repetition count too large");
    ps->pos++;
  }
  *out = (int)v;
  return 1;
}

static int skip_name(parser_t *ps) {
  int n = 0;
  while (!p_eof(ps) && p_peek(ps, 0) != '>') {
    ps->pos++;
    n++;
  }
  if (!n || !p_eat(ps, '>')) return p_fail(ps, "bad capture group name");
  return 1;
}

/* returns NULL with ps->failed == 0 for a flags-only group */
static node_t *parse_group(parser_t *ps, flags_t *f) {
  ps->pos++; /* ( */
  flags_t inner = *f;
  if (p_peek(ps, 0) == '?') {
    uint32_t a = p_peek(ps, 1), b = p_peek(ps, 2);
    if (a == 'P' && b == '<') {
      ps->pos += 3;
      if (!skip_name(ps)) return NULL;
    } else if (a == '<' && b != '=' && b != '!') {
      ps->pos += 2;
      if (!skip_name(ps)) return NULL;
    } else if (a == '=' || a == '!' || (a == '<' && (b == '=' || b == '!'))) {
      p_fail(ps, "look-around is not supported");
      return NULL;
    } else if (a == 'P' && (b == '=' || b == '>')) {
      p_fail(ps, "backreferences are not supported");
      return NULL;
    } else {
      ps->pos++;
      int neg = 0, any = 0;
      for (;;) {
        uint32_t c = p_peek(ps, 0);
        if (c == ':' || c == ')') break;
        if (c == '-') {
          if (neg) {
            p_fail(ps, "dangling flag negation");
            return NULL;
          }
          neg = 1;
          ps->pos++;
          continue;
        }
        any = 1;
        int on = !neg;
        if (c == 'i') inner.i = on;
        else if (c == 's') inner.s = on;
        else if (c == 'x') inner.x = on;
        else if (c == 'U' || c == 'R') { /* greediness / CRLF: no effect on is_match without (?m) */ }
        else if (c == 'm') inner.m = on;
        else if (c == 'u') inner.ascii = !on;
        else { p_fail(ps, "unrecognized flag"); return NULL; }
        ps->pos++;
      }
      if (!any && !neg && p_peek(ps, 0) == ')') {
        p_fail(ps, "missing flags");
        return NULL;
      }
      if (p_eat(ps, ')')) {
        *f = inner;
        return NULL;
      }
      ps->pos++; /* : */
    }
  }
  node_t *body = parse_alt(ps, &inner);
  if (ps->failed) {
    node_free(body);
    return NULL;
  }
  if (!p_eat(ps, ')')) {
    node_free(body);
    p_fail(ps, "unclosed group");
    return NULL;
  }
  return body;
}

static node_t *parse_atom(parser_t *ps, flags_t *f) {
  uint32_t c = p_peek(ps, 0);
  ps->ascii = f->ascii;
  if (c == '(') return parse_group(ps, f);
  if (c == '[') {
    node_t *n = node_new(N_CLASS);
    if (!parse_class(ps, f, &n->cls)) {
      node_free(n);
      return NULL;
    }
    return n;
  }
  if (c == '.' && f->ascii) {
    p_fail(ps, "`.` under (?-u) can match invalid UTF-8");
    return NULL;
  }
  if (c == '.') {
    ps->pos++;
    node_t *n = node_new(N_CLASS);
    if (f->s) {
      cls_add(&n->cls, 0, MAXCP);
    } else {
      cls_add(&n->cls, '\n', '\n');
      cls_negate(&n->cls);
    }
    return n;
  }
  if (c == '^') {
    ps->pos++;
    return node_new(f->m ? N_BOL : N_START);
  }
  if (c == '$') {
    ps->pos++;
    return node_new(f->m ? N_EOL : N_END);
  }
  if (c == '*' || c == '+' || c == '?' || c == '{') {
    p_fail(ps, "repetition operator missing expression");
    return NULL;
  }
  if (c == '\\') {
    ps->pos++;
    if (p_eof(ps)) {
      p_fail(ps, "incomplete escape sequence");
      return NULL;
    }
    uint32_t e = ps->p[ps->pos++];
    if (e == 'A') return node_new(N_START);
    if (e == 'z') return node_new(N_END);
    if (e == 'b' || e == 'B') {
      if (!f->ascii) return node_new(e == 'b' ? N_UWORDB : N_UNWORDB); /* \w of the characters on either side */
      return node_new(e == 'b' ? N_WORDB : N_NWORDB);
    }
    if (e >= '1' && e <= '9') {
      p_fail(ps, "backreferences are not supported");
      return NULL;
    }
    node_t *n = node_new(N_CLASS);
    int is_class = 0;
    if (!class_escape(ps, e, &n->cls, &is_class)) {
      node_free(n);
      return NULL;
    }
    if (is_class) {
      if (f->i) cls_fold(&n->cls);
      return n;
    }
    node_free(n);
    uint32_t cp = 0;
    int handled = 0;
    if (!simple_escape(ps, e, &cp, &handled)) return NULL;
    if (!handled) {
      p_fail(ps, "unrecognized escape sequence");
      return NULL;
    }
    return mk_literal(cp, f);
  }
  ps->pos++;
  return mk_literal(c, f);
}

static node_t *parse_concat(parser_t *ps, flags_t *f) {
  node_t *cat = node_new(N_CAT);
  for (;;) {
    skip_ws(ps, f);
    uint32_t c = p_peek(ps, 0);
    if (p_eof(ps) || c == '|' || c == ')') break;
    node_t *atom = parse_atom(ps, f);
    if (ps->failed) {
      node_free(atom);
      node_free(cat);
      return NULL;
    }
    if (!atom) continue;
    for (;;) {
      skip_ws(ps, f);
      uint32_t q = p_peek(ps, 0);
      int mn = 0, mx = 0;
      if (q == '*') { mn = 0; mx = -1; ps->pos++; }
      else if (q == '+') { mn = 1; mx = -1; ps->pos++; }
      else if (q == '?') { mn = 0; mx = 1; ps->pos++; }
      else if (q == '{') {
        ps->pos++;
        while (p_peek(ps, 0) == ' ') ps->pos++;
        int ok = parse_dec(ps, &mn);
        while (ok && p_peek(ps, 0) == ' ') ps->pos++;
        if (ok && p_eat(ps, '}')) {
          mx = mn;
        } else if (ok && p_eat(ps, ',')) {
          while (p_peek(ps, 0) == ' ') ps->pos++;
          if (p_eat(ps, '}')) {
            mx = -1;
          } else {
            ok = parse_dec(ps, &mx);
            while (ok && p_peek(ps, 0) == ' ') ps->pos++;
            ok = ok && p_eat(ps, '}');
            if (ok && mx < mn) {
              p_fail(ps, "invalid repetition count range");
              ok = 0;
            }
          }
        } else {
          ok = 0;
        }
        if (!ok || mn > 1000 || mx > 1000) {
          p_fail(ps, "invalid counted repetition");
          node_free(atom);
          node_free(cat);
          return NULL;
        }
      } else {
        break;
      }
      if (p_peek(ps, 0) == '?') ps->pos++; /* lazy */
      node_t *rep = node_new(N_REP);
      rep->min = mn;
      rep->max = mx;
      node_push(rep, atom);
      atom = rep;
    }
    node_push(cat, atom);
  }
  return cat;
}

static node_t *parse_alt(parser_t *ps, flags_t *f_in) {
  flags_t f = *f_in; /* flags set inside a group die with the group */
  if (++ps->depth > 200) {
    p_fail(ps, "nesting too deep");
    return NULL;
  }
  node_t *alt = node_new(N_ALT);
  for (;;) {
    node_t *c = parse_concat(ps, &f);
    if (ps->failed) {
      node_free(c);
      node_free(alt);
      return NULL;
    }
    node_push(alt, c);
    if (!p_eat(ps, '|')) break;
  }
  ps->depth--;
  return alt;
}

/* ---------------------------------------------------------------- program (Pike VM) */
enum { I_CLASS, I_SPLIT, I_JMP, I_START, I_END, I_BOL, I_EOL, I_WORDB, I_NWORDB, I_UWORDB, I_UNWORDB, I_MATCH };
typedef struct {
  int op;
  int x, y;         /* targets / class index */
} inst_t;

struct orc_regex {
  inst_t *prog;
  int n, cap;
  cls_t *classes;
  int n_classes, cap_classes;
  int too_big;
};

static int emit(orc_regex *re, int op, int x, int y) {
  if (re->n >= 400000) {
    re->too_big = 1;
    return 0;
  }
  if (re->n == re->cap) {
    re->cap = re->cap ? re->cap * 2 : 64;
    re->prog = (inst_t *)realloc(re->prog, (size_t)re->cap * sizeof(inst_t));
  }
  re->prog[re->n].op = op;
  re->prog[re->n].x = x;
  re->prog[re->n].y = y;
  return re->n++;
}

static int add_class(orc_regex *re, const cls_t *c) {
  if (re->n_classes == re->cap_classes) {
    re->cap_classes = re->cap_classes ? re->cap_classes * 2 : 16;
    re->classes = (cls_t *)realloc(re->classes, (size_t)re->cap_classes * sizeof(cls_t));
  }
  cls_t copy = {0, 0, 0};
  cls_union(&copy, c);
  re->classes[re->n_classes] = copy;
  return re->n_classes++;
}

/* emits code for n that falls through to the next instruction on success */
static void gen(orc_regex *re, const node_t *n) {
  switch (n->kind) {
    case N_EMPTY: break;
    case N_CLASS: emit(re, I_CLASS, add_class(re, &n->cls), 0); break;
    case N_START: emit(re, I_START, 0, 0); break;
    case N_END: emit(re, I_END, 0, 0); break;
    case N_BOL: emit(re, I_BOL, 0, 0); break;
    case N_EOL: emit(re, I_EOL, 0, 0); break;
    case N_WORDB: emit(re, I_WORDB, 0, 0); break;
    case N_NWORDB: emit(re, I_NWORDB, 0, 0); break;
    case N_UWORDB: emit(re, I_UWORDB, 0, 0); break;
    case N_UNWORDB: emit(re, I_UNWORDB, 0, 0); break;
    case N_CAT:
      for (int i = 0; i < n->nkid; i++) gen(re, n->kid[i]);
      break;
    case N_ALT: {
      if (n->nkid == 1) {
        gen(re, n->kid[0]);
        break;
      }
      /* split L1, L2; L1: a; jmp end; L2: split ... */
      int *jmps = (int *)malloc((size_t)n->nkid * sizeof(int));
      for (int i = 0; i < n->nkid; i++) {
        int sp = -1;
        if (i + 1 < n->nkid) sp = emit(re, I_SPLIT, 0, 0);
        if (sp >= 0) re->prog[sp].x = re->n;
        gen(re, n->kid[i]);
        jmps[i] = (i + 1 < n->nkid) ? emit(re, I_JMP, 0, 0) : -1;
        if (sp >= 0) re->prog[sp].y = re->n;
      }
      for (int i = 0; i + 1 < n->nkid; i++) re->prog[jmps[i]].x = re->n;
      free(jmps);
      break;
    }
    case N_REP: {
      const node_t *body = n->kid[0];
      for (int k = 0; k < n->min; k++) gen(re, body);
      if (n->max < 0) {
        /* L: split B, end; B: body; jmp L */
        int sp = emit(re, I_SPLIT, 0, 0);
        re->prog[sp].x = re->n;
        gen(re, body);
        emit(re, I_JMP, sp, 0);
        re->prog[sp].y = re->n;
      } else {
        int cnt = n->max - n->min;
        int *sps = (int *)malloc((size_t)(cnt > 0 ? cnt : 1) * sizeof(int));
        for (int k = 0; k < cnt; k++) {
          sps[k] = emit(re, I_SPLIT, 0, 0);
          re->prog[sps[k]].x = re->n;
          gen(re, body);
        }
        for (int k = 0; k < cnt; k++) re->prog[sps[k]].y = re->n;
        free(sps);
      }
      break;
    }
  }
}

static int decode_utf8(const uint8_t *s, size_t n, uint32_t **out, int *count) {
  uint32_t *cp = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
  int m = 0;
  size_t i = 0;
  while (i < n) {
    uint8_t b = s[i];
    uint32_t c;
    int len;
    if (b < 0x80) { c = b; len = 1; }
    else if ((b & 0xE0) == 0xC0) { c = b & 0x1F; len = 2; }
    else if ((b & 0xF0) == 0xE0) { c = b & 0x0F; len = 3; }
    else if ((b & 0xF8) == 0xF0) { c = b & 0x07; len = 4; }
    else { free(cp); return 0; }
    if (i + (size_t)len > n) { free(cp); return 0; }
    for (int k = 1; k < len; k++) c = (c << 6) | (s[i + (size_t)k] & 0x3F);
    cp[m++] = c;
    i += (size_t)len;
  }
  *out = cp;
  *count = m;
  return 1;
}

orc_regex *orc_regex_compile(const char *pattern, size_t len, int case_insensitive, char *err, size_t cap) {
  parser_t ps;
  memset(&ps, 0, sizeof(ps));
  if (!decode_utf8((const uint8_t *)pattern, len, &ps.p, &ps.n)) {
    if (err) snprintf(err, cap, "pattern is not valid UTF-8");
    return NULL;
  }
  flags_t f = {case_insensitive ? 1 : 0, 0, 0};
  node_t *ast = parse_alt(&ps, &f);
  if (!ps.failed && !p_eof(&ps)) p_fail(&ps, p_peek(&ps, 0) == ')' ? "unopened group" : "unexpected character");
  if (ps.failed) {
    if (err) snprintf(err, cap, "%s", ps.msg);
    node_free(ast);
    free(ps.p);
    return NULL;
  }
  orc_regex *re = (orc_regex *)calloc(1, sizeof(orc_regex));
  gen(re, ast);
  emit(re, I_MATCH, 0, 0);
  node_free(ast);
  free(ps.p);
  if (re->too_big) {
    if (err) snprintf(err, cap, "program too large");
    orc_regex_free(re);
    return NULL;
  }
  return re;
}

void orc_regex_free(orc_regex *re) {
  if (!re) return;
  for (int i = 0; i < re->n_classes; i++) free(re->classes[i].r);
  free(re->classes);
  free(re->prog);
  free(re);
}

typedef struct {
  int *pc;
  int n;
} tlist_t;

/* follow empty-width instructions from pc; mark[] prevents revisits within one step */
static int ascii_word(long cp) {
  return (cp >= '0' && cp <= '9') || (cp >= 'A' && cp <= 'Z') || cp == '_' || (cp >= 'a' && cp <= 'z');
}
static int uni_word(long cp) { /* \w under Unicode: the same table the class escape uses */
  static cls_t word;
  static int built = 0;
  if (!built) {
    memset(&word, 0, sizeof(word));
    cls_table(&word, "perl_word");
    built = 1;
  }
  return cp >= 0 && cls_has(&word, (uint32_t)cp);
}
/* prev / next: the code points on either side of the position, -1 at the ends of the haystack */
static int add_thread(const orc_regex *re, tlist_t *l, int *mark, int gen_id, int pc, long prev, long next) {
  const int at_start = prev < 0, at_end = next < 0;
  /* iterative DFS; a pc is marked when pushed, so the stack never holds more than re->n entries */
  int *stack = (int *)malloc((size_t)(re->n + 1) * sizeof(int));
  int sp = 0, matched = 0;
#define PUSH(q)                 \
  do {                          \
    if (mark[(q)] != gen_id) {  \
      mark[(q)] = gen_id;       \
      stack[sp++] = (q);        \
    }                           \
  } while (0)
  PUSH(pc);
  while (sp) {
    int p = stack[--sp];
    const inst_t *in = &re->prog[p];
    switch (in->op) {
      case I_JMP: PUSH(in->x); break;
      case I_SPLIT:
        PUSH(in->y);
        PUSH(in->x);
        break;
      case I_START:
        if (at_start) PUSH(p + 1);
        break;
      case I_END:
        if (at_end) PUSH(p + 1);
        break;
      case I_BOL: /* (?m)^ : at the start or after a line feed */
        if (at_start || prev == '\n') PUSH(p + 1);
        break;
      case I_EOL: /* (?m)$ : at the end or before a line feed */
        if (at_end || next == '\n') PUSH(p + 1);
        break;
      case I_WORDB: /* (?-u:\b) */
        if (ascii_word(prev) != ascii_word(next)) PUSH(p + 1);
        break;
      case I_NWORDB:
        if (ascii_word(prev) == ascii_word(next)) PUSH(p + 1);
        break;
      case I_UWORDB: /* \b: exactly one of the two neighbours is a \w character (the ends of the haystack are not) */
        if (uni_word(prev) != uni_word(next)) PUSH(p + 1);
        break;
      case I_UNWORDB:
        if (uni_word(prev) == uni_word(next)) PUSH(p + 1);
        break;
      case I_MATCH: matched = 1; break;
      default: l->pc[l->n++] = p; break;
    }
  }
#undef PUSH
  free(stack);
  return matched;
}

int orc_regex_is_match(const orc_regex *re, const uint8_t *s, size_t len) {
  uint32_t *cp = NULL;
  int n = 0;
  if (!decode_utf8(s, len, &cp, &n)) return 0;
  tlist_t cur = {(int *)malloc((size_t)(re->n + 1) * sizeof(int)), 0};
  tlist_t nxt = {(int *)malloc((size_t)(re->n + 1) * sizeof(int)), 0};
  int *mark = (int *)calloc((size_t)re->n + 1, sizeof(int));
  int gen_id = 0, matched = 0;
  for (int i = 0; i <= n && !matched; i++) {
    /* threads alive at position i: survivors (already in cur) + a fresh one (unanchored search) */
    gen_id++;
    /* re-close survivors is unnecessary: they were closed when added; but a fresh thread must not
       duplicate them, so mark the survivors first */
    for (int k = 0; k < cur.n; k++) mark[cur.pc[k]] = gen_id;
    const long prev = i > 0 ? (long)cp[i - 1] : -1, next = i < n ? (long)cp[i] : -1;
    if (add_thread(re, &cur, mark, gen_id, 0, prev, next)) matched = 1;
    if (matched || i == n) break;
    /* step over cp[i] */
    nxt.n = 0;
    gen_id++;
    for (int k = 0; k < cur.n && !matched; k++) {
      const inst_t *in = &re->prog[cur.pc[k]];
      if (in->op == I_CLASS && cls_has(&re->classes[in->x], cp[i]))
        if (add_thread(re, &nxt, mark, gen_id, cur.pc[k] + 1, (long)cp[i], i + 1 < n ? (long)cp[i + 1] : -1)) matched = 1;
    }
    tlist_t t = cur;
    cur = nxt;
    nxt = t;
  }
  free(cur.pc);
  free(nxt.pc);
  free(mark);
  free(cp);
  return matched;
}

static inline int bit_set(const uint8_t *bm, int64_t i) { return bm == NULL ? 1 : (bm[i >> 3] >> (i & 7)) & 1; }

/* TG/constraints/format.rs:762-776: matches = rows where ([TRIM(]c[)] ~ pat) [OR c IS NULL]; total = COUNT(*).
 * TRIM is SQL btrim with the default character set: U+0020 only. */
void orc_regex_count_utf8(const orc_regex *re, const int32_t *offsets, const uint8_t *data,
                          const uint8_t *validity, int64_t offset, int64_t n, int trim, int null_is_valid,
                          orc_match_t *out) {
  out->total = n;
  out->matches = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!bit_set(validity, offset + i)) {
      out->matches += null_is_valid ? 1 : 0;
      continue;
    }
    int64_t b = offsets[offset + i], e = offsets[offset + i + 1];
    if (trim) {
      while (b < e && data[b] == 0x20) b++;
      while (e > b && data[e - 1] == 0x20) e--;
    }
    out->matches += orc_regex_is_match(re, data + b, (size_t)(e - b));
  }
}

/* TG/constraints/length.rs:36-45, 167-171: matches = rows where LENGTH(c) in [min_chars, max_chars] OR c IS NULL;
 * total = COUNT(*).  LENGTH counts characters: bytes that do not continue a UTF-8 sequence (10xxxxxx). */
void orc_length_count_utf8(const int32_t *offsets, const uint8_t *data, const uint8_t *validity, int64_t offset,
                           int64_t n, uint64_t min_chars, uint64_t max_chars, orc_match_t *out) {
  out->total = n;
  out->matches = 0;
  for (int64_t i = 0; i < n; i++) {
    if (!bit_set(validity, offset + i)) {
      out->matches += 1;
      continue;
    }
    uint64_t chars = 0;
    for (int64_t p = offsets[offset + i]; p < offsets[offset + i + 1]; p++) chars += (data[p] & 0xC0) != 0x80;
    out->matches += chars >= min_chars && chars <= max_chars;
  }
}

