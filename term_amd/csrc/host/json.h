// json.h -- a small JSON value + parser + writer for the suite bridge (no external dependency).
#pragma once
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace term_guard {
namespace json {

struct Value {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0;
  // A Number whose token was an integer keeps it exactly: serde_json reads a `u64` / `i64` field only from an integer
  // token (TG/analyzers/incremental/runner.rs:86,98), and a count above 2^53 does not fit a double.  `num` always holds
  // the nearest double as well, so readers of f64 fields need not care.
  enum IntKind { NotInt, Signed, Unsigned } int_kind = NotInt;
  int64_t i = 0;    // int_kind == Signed (negative values)
  uint64_t u = 0;   // int_kind == Unsigned (every non-negative integer token)
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;  // insertion ordered

  bool is(Type t) const { return type == t; }
  const Value *get(const std::string &key) const {
    if (type != Object) return nullptr;
    for (auto &kv : obj)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  std::string get_str(const std::string &key, const std::string &dflt = "") const {
    const Value *v = get(key);
    return v && v->type == String ? v->str : dflt;
  }
  double get_num(const std::string &key, double dflt = 0) const {
    const Value *v = get(key);
    return v && v->type == Number ? v->num : dflt;
  }
  // an integer field: exact from an integer token; a float token (a state written by an older build or by a Python
  // float) is accepted when it is integral and in range
  uint64_t as_u64() const {
    if (type != Number) return 0;
    if (int_kind == Unsigned) return u;
    if (int_kind == Signed) return i < 0 ? 0 : (uint64_t)i;
    if (!(num >= 0)) return 0;
    return num >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)num;
  }
  int64_t as_i64() const {
    if (type != Number) return 0;
    if (int_kind == Signed) return i;
    if (int_kind == Unsigned) return u > (uint64_t)INT64_MAX ? INT64_MAX : (int64_t)u;
    if (num != num) return 0;
    return num >= 9223372036854775807.0 ? INT64_MAX : num <= -9223372036854775808.0 ? INT64_MIN : (int64_t)num;
  }
  uint64_t get_u64(const std::string &key, uint64_t dflt = 0) const {
    const Value *v = get(key);
    return v && v->type == Number ? v->as_u64() : dflt;
  }
  int64_t get_i64(const std::string &key, int64_t dflt = 0) const {
    const Value *v = get(key);
    return v && v->type == Number ? v->as_i64() : dflt;
  }
  static Value of_u64(uint64_t x) {
    Value v;
    v.type = Number;
    v.int_kind = Unsigned;
    v.u = x;
    v.num = (double)x;
    return v;
  }
  static Value of_i64(int64_t x) {
    if (x >= 0) return of_u64((uint64_t)x);
    Value v;
    v.type = Number;
    v.int_kind = Signed;
    v.i = x;
    v.num = (double)x;
    return v;
  }
  bool get_bool(const std::string &key, bool dflt = false) const {
    const Value *v = get(key);
    return v && v->type == Bool ? v->b : dflt;
  }
};

struct Parser {
  const char *p, *e;
  std::string err;
  explicit Parser(const std::string &s) : p(s.data()), e(s.data() + s.size()) {}
  void ws() {
    while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
  }
  bool fail(const char *m) {
    if (err.empty()) err = m;
    return false;
  }
  static void put_utf8(std::string &out, unsigned cp) {
    if (cp < 0x80) {
      out.push_back((char)cp);
    } else if (cp < 0x800) {
      out.push_back((char)(0xC0 | (cp >> 6)));
      out.push_back((char)(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
      out.push_back((char)(0xE0 | (cp >> 12)));
      out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
      out.push_back((char)(0xF0 | (cp >> 18)));
      out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
      out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back((char)(0x80 | (cp & 0x3F)));
    }
  }
  bool hex4(unsigned *out) {
    if (e - p < 4) return fail("bad \\u escape");
    unsigned v = 0;
    for (int i = 0; i < 4; i++) {
      char c = p[i];
      int h = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
      if (h < 0) return fail("bad \\u escape");
      v = v * 16 + (unsigned)h;
    }
    p += 4;
    *out = v;
    return true;
  }
  bool string(std::string *out) {
    if (p >= e || *p != '"') return fail("expected string");
    p++;
    while (p < e && *p != '"') {
      if (*p == '\\') {
        p++;
        if (p >= e) return fail("bad escape");
        char c = *p++;
        switch (c) {
          case 'n': out->push_back('\n'); break;
          case 't': out->push_back('\t'); break;
          case 'r': out->push_back('\r'); break;
          case 'b': out->push_back('\b'); break;
          case 'f': out->push_back('\f'); break;
          case 'u': {
            unsigned cp;
            if (!hex4(&cp)) return false;
            if (cp >= 0xD800 && cp <= 0xDBFF && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {
              p += 2;
              unsigned lo;
              if (!hex4(&lo)) return false;
              cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
            }
            put_utf8(*out, cp);
            break;
          }
          default: out->push_back(c); break;
        }
      } else {
        out->push_back(*p++);
      }
    }
    if (p >= e) return fail("unterminated string");
    p++;
    return true;
  }
  bool value(Value *v) {
    ws();
    if (p >= e) return fail("unexpected end");
    char c = *p;
    if (c == '{') {
      p++;
      v->type = Value::Object;
      ws();
      if (p < e && *p == '}') {
        p++;
        return true;
      }
      for (;;) {
        ws();
        std::string k;
        if (!string(&k)) return false;
        ws();
        if (p >= e || *p != ':') return fail("expected ':'");
        p++;
        Value child;
        if (!value(&child)) return false;
        v->obj.emplace_back(std::move(k), std::move(child));
        ws();
        if (p < e && *p == ',') {
          p++;
          continue;
        }
        if (p < e && *p == '}') {
          p++;
          return true;
        }
        return fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      p++;
      v->type = Value::Array;
      ws();
      if (p < e && *p == ']') {
        p++;
        return true;
      }
      for (;;) {
        Value child;
        if (!value(&child)) return false;
        v->arr.push_back(std::move(child));
        ws();
        if (p < e && *p == ',') {
          p++;
          continue;
        }
        if (p < e && *p == ']') {
          p++;
          return true;
        }
        return fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      v->type = Value::String;
      return string(&v->str);
    }
    if (e - p >= 4 && std::string(p, 4) == "true") {
      p += 4;
      v->type = Value::Bool;
      v->b = true;
      return true;
    }
    if (e - p >= 5 && std::string(p, 5) == "false") {
      p += 5;
      v->type = Value::Bool;
      v->b = false;
      return true;
    }
    if (e - p >= 4 && std::string(p, 4) == "null") {
      p += 4;
      v->type = Value::Null;
      return true;
    }
    // numbers; NaN / Infinity / -Infinity accepted as Python's json emits them
    if (e - p >= 3 && std::string(p, 3) == "NaN") {
      p += 3;
      v->type = Value::Number;
      v->num = NAN;
      return true;
    }
    if (e - p >= 8 && std::string(p, 8) == "Infinity") {
      p += 8;
      v->type = Value::Number;
      v->num = INFINITY;
      return true;
    }
    if (e - p >= 9 && std::string(p, 9) == "-Infinity") {
      p += 9;
      v->type = Value::Number;
      v->num = -INFINITY;
      return true;
    }
    char *end = nullptr;
    std::string tmp(p, (size_t)std::min<ptrdiff_t>(e - p, 64));
    double d = strtod(tmp.c_str(), &end);
    if (end == tmp.c_str()) return fail("unexpected token");
    const size_t len = (size_t)(end - tmp.c_str());
    p += len;
    v->type = Value::Number;
    v->num = d;
    // an integer token (no fraction, no exponent) that fits 64 bits is kept exactly
    bool integral = true;
    for (size_t k = 0; k < len; k++) {
      const char ch = tmp[k];
      if (!((ch >= '0' && ch <= '9') || (k == 0 && ch == '-'))) integral = false;
    }
    if (integral && len > (tmp[0] == '-' ? 1u : 0u)) {
      errno = 0;
      if (tmp[0] == '-') {
        const long long sv = strtoll(tmp.c_str(), nullptr, 10);
        if (errno == 0) {
          if (sv == 0) {
            *v = Value::of_u64(0);
            v->num = d;  // "-0" stays -0.0 for an f64 reader
          } else {
            *v = Value::of_i64((int64_t)sv);
          }
        }
      } else {
        const unsigned long long uv = strtoull(tmp.c_str(), nullptr, 10);
        if (errno == 0) *v = Value::of_u64((uint64_t)uv);
      }
    }
    return true;
  }
};

inline bool parse(const std::string &text, Value *out, std::string *err) {
  Parser ps(text);
  if (!ps.value(out)) {
    *err = ps.err;
    return false;
  }
  ps.ws();
  if (ps.p != ps.e) {
    *err = "trailing characters";
    return false;
  }
  return true;
}

inline std::string quote(const std::string &s) {
  std::string o = "\"";
  for (unsigned char c : s) {
    switch (c) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\n': o += "\\n"; break;
      case '\t': o += "\\t"; break;
      case '\r': o += "\\r"; break;
      default:
        if (c < 0x20) {
          char buf[8];
          snprintf(buf, sizeof(buf), "\\u%04x", c);
          o += buf;
        } else {
          o.push_back((char)c);
        }
    }
  }
  o += "\"";
  return o;
}

}  // namespace json
}  // namespace term_guard
