// json_min.h -- a small JSON reader (objects, arrays, strings, numbers, true/false/null), enough for glTF.
#ifndef RRT_JSON_MIN_H
#define RRT_JSON_MIN_H

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rrtjson {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<ValuePtr> arr;
    std::vector<std::pair<std::string, ValuePtr>> obj;  // insertion order kept

    const Value* get(const std::string& key) const {
        if (type != Object) return nullptr;
        for (const auto& kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    bool is_number() const { return type == Number; }
    bool is_string() const { return type == String; }
    bool is_array() const { return type == Array; }
    bool is_object() const { return type == Object; }
    bool is_bool() const { return type == Bool; }
    size_t size() const { return type == Array ? arr.size() : 0; }
    // bounds-checked: indices often come straight from the file being read
    const Value& at(size_t i) const {
        if (type != Array || i >= arr.size() || !arr[i]) throw std::out_of_range("JSON array index out of range");
        return *arr[i];
    }
};

class Parser {
  public:
    Parser(const std::string& text) : s_(text) {}
    // returns nullptr and sets error() on failure
    ValuePtr parse() {
        pos_ = 0;
        ValuePtr v = value(0);
        if (!v) return nullptr;
        ws();
        if (pos_ != s_.size()) return fail("trailing characters after JSON value");
        return v;
    }
    const std::string& error() const { return err_; }

  private:
    const std::string& s_;
    size_t pos_ = 0;
    std::string err_;

    ValuePtr fail(const std::string& m) {
        if (err_.empty()) err_ = m + " at byte " + std::to_string(pos_);
        return nullptr;
    }
    void ws() {
        while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\t' || s_[pos_] == '\n' || s_[pos_] == '\r')) ++pos_;
    }
    bool lit(const char* w) {
        size_t n = 0;
        while (w[n]) ++n;
        if (s_.compare(pos_, n, w) == 0) { pos_ += n; return true; }
        return false;
    }
    static void put_utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    bool string(std::string& out) {
        if (pos_ >= s_.size() || s_[pos_] != '"') { fail("expected string"); return false; }
        ++pos_;
        while (pos_ < s_.size()) {
            char c = s_[pos_++];
            if (c == '"') return true;
            if (c == '\\') {
                if (pos_ >= s_.size()) break;
                char e = s_[pos_++];
                switch (e) {
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case '/': out += '/'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'u': {
                        if (pos_ + 4 > s_.size()) { fail("bad \\u escape"); return false; }
                        unsigned cp = (unsigned)std::strtoul(s_.substr(pos_, 4).c_str(), nullptr, 16);
                        pos_ += 4;
                        put_utf8(out, cp);
                        break;
                    }
                    default: fail("bad escape"); return false;
                }
            } else {
                out += c;
            }
        }
        fail("unterminated string");
        return false;
    }
    ValuePtr value(int depth) {
        if (depth > 256) return fail("JSON nested too deeply");
        ws();
        if (pos_ >= s_.size()) return fail("unexpected end of JSON");
        auto v = std::make_shared<Value>();
        char c = s_[pos_];
        if (c == '{') {
            v->type = Value::Object;
            ++pos_;
            ws();
            if (pos_ < s_.size() && s_[pos_] == '}') { ++pos_; return v; }
            for (;;) {
                ws();
                std::string key;
                if (!string(key)) return nullptr;
                ws();
                if (pos_ >= s_.size() || s_[pos_] != ':') return fail("expected ':'");
                ++pos_;
                ValuePtr m = value(depth + 1);
                if (!m) return nullptr;
                v->obj.emplace_back(std::move(key), m);
                ws();
                if (pos_ < s_.size() && s_[pos_] == ',') { ++pos_; continue; }
                if (pos_ < s_.size() && s_[pos_] == '}') { ++pos_; return v; }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v->type = Value::Array;
            ++pos_;
            ws();
            if (pos_ < s_.size() && s_[pos_] == ']') { ++pos_; return v; }
            for (;;) {
                ValuePtr m = value(depth + 1);
                if (!m) return nullptr;
                v->arr.push_back(m);
                ws();
                if (pos_ < s_.size() && s_[pos_] == ',') { ++pos_; continue; }
                if (pos_ < s_.size() && s_[pos_] == ']') { ++pos_; return v; }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v->type = Value::String;
            if (!string(v->str)) return nullptr;
            return v;
        }
        if (lit("true")) { v->type = Value::Bool; v->b = true; return v; }
        if (lit("false")) { v->type = Value::Bool; v->b = false; return v; }
        if (lit("null")) { v->type = Value::Null; return v; }
        if (c == '-' || (c >= '0' && c <= '9')) {
            const char* start = s_.c_str() + pos_;
            char* end = nullptr;
            v->num = std::strtod(start, &end);
            if (end == start) return fail("bad number");
            pos_ += (size_t)(end - start);
            v->type = Value::Number;
            return v;
        }
        return fail("unexpected character");
    }
};

}  // namespace rrtjson
#endif
