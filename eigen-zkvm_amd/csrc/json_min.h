// Minimal JSON reader for the prover's inputs (serialised StarkInfo / Program / StarkStruct).
// Numbers keep their source text (u64 values would not survive a double).  Not a general-purpose
// library: no \u escapes beyond pass-through, no streaming.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace zk {

struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    std::string s;  // Num: source text; Str: value
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;

    const JVal* find(const std::string& k) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    const JVal& at(const std::string& k) const {
        const JVal* v = find(k);
        if (!v) throw std::runtime_error("json: missing key '" + k + "'");
        return *v;
    }
    const JVal& at(size_t i) const {
        if (kind != Arr || i >= arr.size()) throw std::runtime_error("json: index out of range");
        return arr[i];
    }
    size_t size() const { return kind == Arr ? arr.size() : kind == Obj ? obj.size() : 0; }
    uint64_t u64() const {
        if (kind != Num) throw std::runtime_error("json: number expected");
        return strtoull(s.c_str(), nullptr, 10);
    }
    int64_t i64() const {
        if (kind != Num) throw std::runtime_error("json: number expected");
        return strtoll(s.c_str(), nullptr, 10);
    }
    const std::string& str() const {
        if (kind != Str) throw std::runtime_error("json: string expected");
        return s;
    }
    bool boolean() const {
        if (kind != Bool) throw std::runtime_error("json: bool expected");
        return b;
    }
    bool is_null() const { return kind == Null; }
};

class JParser {
    const char* p; const char* end;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("json: ") + m); }
    std::string string_() {
        if (*p != '"') fail("'\"' expected");
        ++p;
        std::string o;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                ++p; if (p >= end) fail("bad escape");
                switch (*p) {
                    case 'n': o += '\n'; break; case 't': o += '\t'; break; case 'r': o += '\r'; break;
                    case 'b': o += '\b'; break; case 'f': o += '\f'; break;
                    case 'u': o += "\\u"; break;  // kept verbatim; identifiers in these files are ASCII
                    default: o += *p;
                }
                ++p;
            } else o += *p++;
        }
        if (p >= end) fail("unterminated string");
        ++p;
        return o;
    }
    JVal value() {
        ws();
        if (p >= end) fail("unexpected end");
        JVal v;
        if (*p == '{') {
            v.kind = JVal::Obj; ++p; ws();
            if (*p == '}') { ++p; return v; }
            for (;;) {
                ws(); std::string k = string_(); ws();
                if (*p != ':') fail("':' expected");
                ++p;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (*p == ',') { ++p; continue; }
                if (*p == '}') { ++p; break; }
                fail("',' or '}' expected");
            }
        } else if (*p == '[') {
            v.kind = JVal::Arr; ++p; ws();
            if (*p == ']') { ++p; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (*p == ',') { ++p; continue; }
                if (*p == ']') { ++p; break; }
                fail("',' or ']' expected");
            }
        } else if (*p == '"') {
            v.kind = JVal::Str; v.s = string_();
        } else if (end - p >= 4 && std::string(p, 4) == "true") { v.kind = JVal::Bool; v.b = true; p += 4; }
        else if (end - p >= 5 && std::string(p, 5) == "false") { v.kind = JVal::Bool; v.b = false; p += 5; }
        else if (end - p >= 4 && std::string(p, 4) == "null") { v.kind = JVal::Null; p += 4; }
        else {
            const char* q = p;
            while (p < end && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' || (*p >= '0' && *p <= '9'))) ++p;
            if (p == q) fail("value expected");
            v.kind = JVal::Num; v.s.assign(q, p);
        }
        return v;
    }
public:
    static JVal parse(const char* text) {
        JParser P; P.p = text; P.end = text + std::string::traits_type::length(text);
        JVal v = P.value();
        P.ws();
        if (P.p != P.end) P.fail("trailing characters");
        return v;
    }
};

}  // namespace zk
