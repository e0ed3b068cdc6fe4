// json.h -- a small JSON value + parser + writer for the suite bridge (no external dependency).
#pragma once
#include <math.h>
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
    p += end - tmp.c_str();
    v->type = Value::Number;
    v->num = d;
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
