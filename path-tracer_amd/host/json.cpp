#include "json.hpp"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

class json_parser {
public:
    explicit json_parser(const std::string& t) : s(t.data()), end(t.data() + t.size()), begin(t.data()) {}

    json document() {
        if (end - s >= 3 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB && (unsigned char)s[2] == 0xBF) s += 3;  // BOM
        json v = value(0);
        skip_ws();
        if (s != end) fail("trailing characters after the document");
        return v;
    }

private:
    const char* s;
    const char* end;
    const char* begin;
    static constexpr int kMaxDepth = 256;

    [[noreturn]] void fail(const char* what) const {
        char buf[160];
        std::snprintf(buf, sizeof buf, "json: %s at byte %zu", what, (size_t)(s - begin));
        throw json_error(buf);
    }
    void skip_ws() { while (s != end && (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\r')) s++; }
    bool literal(const char* word) {
        const size_t n = std::strlen(word);
        if ((size_t)(end - s) >= n && std::memcmp(s, word, n) == 0) { s += n; return true; }
        return false;
    }

    json value(int depth) {
        if (depth > kMaxDepth) fail("nesting too deep");
        skip_ws();
        if (s == end) fail("unexpected end of input");
        json v;
        switch (*s) {
        case '{': {
            s++;
            v.k = json::kind::object;
            skip_ws();
            if (s != end && *s == '}') { s++; return v; }
            for (;;) {
                skip_ws();
                if (s == end || *s != '"') fail("object key must be a string");
                std::string key = string();
                skip_ws();
                if (s == end || *s != ':') fail("':' expected");
                s++;
                v.obj.emplace_back(std::move(key), value(depth + 1));
                skip_ws();
                if (s == end) fail("unterminated object");
                if (*s == ',') { s++; continue; }
                if (*s == '}') { s++; return v; }
                fail("',' or '}' expected");
            }
        }
        case '[': {
            s++;
            v.k = json::kind::array;
            skip_ws();
            if (s != end && *s == ']') { s++; return v; }
            for (;;) {
                v.arr.push_back(value(depth + 1));
                skip_ws();
                if (s == end) fail("unterminated array");
                if (*s == ',') { s++; continue; }
                if (*s == ']') { s++; return v; }
                fail("',' or ']' expected");
            }
        }
        case '"':
            v.k = json::kind::string;
            v.str = string();
            return v;
        case 't':
            if (!literal("true")) fail("invalid literal");
            v.k = json::kind::boolean; v.b = true;
            return v;
        case 'f':
            if (!literal("false")) fail("invalid literal");
            v.k = json::kind::boolean; v.b = false;
            return v;
        case 'n':
            if (!literal("null")) fail("invalid literal");
            return v;
        default:
            v.k = json::kind::number;
            v.num = number();
            return v;
        }
    }

    double number() {
        // validate the RFC 8259 grammar first (strtod alone would accept hex, "inf", leading '+', ...)
        const char* p = s;
        if (p != end && *p == '-') p++;
        if (p == end) fail("invalid number");
        if (*p == '0') p++;
        else if (*p >= '1' && *p <= '9') { while (p != end && *p >= '0' && *p <= '9') p++; }
        else fail("invalid number");
        if (p != end && *p == '.') {
            p++;
            if (p == end || *p < '0' || *p > '9') fail("digit expected after '.'");
            while (p != end && *p >= '0' && *p <= '9') p++;
        }
        if (p != end && (*p == 'e' || *p == 'E')) {
            p++;
            if (p != end && (*p == '+' || *p == '-')) p++;
            if (p == end || *p < '0' || *p > '9') fail("digit expected in exponent");
            while (p != end && *p >= '0' && *p <= '9') p++;
        }
        const std::string token(s, p);  // NUL-terminated copy for strtod (correctly rounded to double)
        s = p;
        return std::strtod(token.c_str(), nullptr);
    }

    static void append_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    uint32_t hex4() {
        if (end - s < 4) fail("truncated \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = *s++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("invalid \\u escape");
        }
        return v;
    }
    std::string string() {
        s++;  // opening quote
        std::string out;
        for (;;) {
            if (s == end) fail("unterminated string");
            const unsigned char c = (unsigned char)*s++;
            if (c == '"') return out;
            if (c < 0x20) fail("control character in string");
            if (c != '\\') { out.push_back((char)c); continue; }
            if (s == end) fail("unterminated escape");
            const char e = *s++;
            switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                uint32_t cp = hex4();
                if (cp >= 0xD800 && cp <= 0xDBFF) {  // surrogate pair
                    if (end - s < 2 || s[0] != '\\' || s[1] != 'u') fail("lone high surrogate");
                    s += 2;
                    const uint32_t lo = hex4();
                    if (lo < 0xDC00 || lo > 0xDFFF) fail("invalid low surrogate");
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) fail("lone low surrogate");
                append_utf8(out, cp);
                break;
            }
            default: fail("invalid escape");
            }
        }
    }
};

json json::parse(const std::string& text) { return json_parser(text).document(); }

json json::parse_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw json_error("json: cannot open " + path);
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
    return parse(text);
}

bool json::contains(const std::string& key) const {
    if (k != kind::object) return false;
    for (const auto& m : obj) if (m.first == key) return true;
    return false;
}
const json& json::operator[](const std::string& key) const {
    if (k != kind::object) throw json_error("json: [\"" + key + "\"] on a value that is not an object");
    for (size_t i = obj.size(); i-- > 0;) if (obj[i].first == key) return obj[i].second;
    throw json_error("json: key \"" + key + "\" is missing");
}
const json& json::operator[](size_t index) const {
    if (k != kind::array) throw json_error("json: [index] on a value that is not an array");
    if (index >= arr.size()) throw json_error("json: array index " + std::to_string(index) + " out of range (size " + std::to_string(arr.size()) + ")");
    return arr[index];
}
size_t json::size() const { return k == kind::array ? arr.size() : k == kind::object ? obj.size() : 0; }
double json::as_double() const {
    if (k != kind::number) throw json_error("json: number expected");
    return num;
}
int64_t json::as_int() const {
    const double d = as_double();
    if (d != std::floor(d) || std::fabs(d) > 9007199254740992.0) throw json_error("json: integer expected");
    return (int64_t)d;
}
uint32_t json::as_u32() const {
    const int64_t v = as_int();
    if (v < 0 || v > 0xFFFFFFFFll) throw json_error("json: unsigned 32-bit integer expected");
    return (uint32_t)v;
}
bool json::as_bool() const {
    if (k != kind::boolean) throw json_error("json: boolean expected");
    return b;
}
const std::string& json::as_string() const {
    if (k != kind::string) throw json_error("json: string expected");
    return str;
}

std::string json::dump() const {
    switch (k) {
    case kind::null: return "null";
    case kind::boolean: return b ? "true" : "false";
    case kind::number: {
        char buf[40];
        std::snprintf(buf, sizeof buf, "%.17g", num);
        return buf;
    }
    case kind::string: {
        std::string out = "\"";
        for (const unsigned char c : str) {
            if (c == '"') out += "\\\"";
            else if (c == '\\') out += "\\\\";
            else if (c < 0x20) { char buf[8]; std::snprintf(buf, sizeof buf, "\\u%04x", c); out += buf; }
            else out.push_back((char)c);
        }
        return out + "\"";
    }
    case kind::array: {
        std::string out = "[";
        for (size_t i = 0; i < arr.size(); i++) { if (i) out += ","; out += arr[i].dump(); }
        return out + "]";
    }
    default: {
        std::string out = "{";
        for (size_t i = 0; i < obj.size(); i++) {
            if (i) out += ",";
            json key; key.k = kind::string; key.str = obj[i].first;
            out += key.dump() + ":" + obj[i].second.dump();
        }
        return out + "}";
    }
    }
}
