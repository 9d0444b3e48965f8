// Minimal JSON reader (RFC 8259) for glTF files: a DOM of null / bool / number / string / array / object.
// The reference parses glTF with nlohmann::json (include/gltf.hpp:6-8); only the access patterns src/gltf.cpp
// uses are offered: operator[] by key and by index, contains(), size(), typed getters.
#ifndef PTR_JSON_HPP_
#define PTR_JSON_HPP_

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

struct json_error : std::runtime_error { using std::runtime_error::runtime_error; };

class json {
public:
    enum class kind { null, boolean, number, string, array, object };

    json() = default;
    static json parse(const std::string& text);
    static json parse_file(const std::string& path);

    kind type() const { return k; }
    bool is_null() const { return k == kind::null; }
    bool is_number() const { return k == kind::number; }
    bool is_string() const { return k == kind::string; }
    bool is_array() const { return k == kind::array; }
    bool is_object() const { return k == kind::object; }

    // objects
    bool contains(const std::string& key) const;
    const json& operator[](const std::string& key) const;  // throws json_error when absent
    const json& operator[](const char* key) const { return (*this)[std::string(key)]; }
    const std::vector<std::pair<std::string, json>>& members() const { return obj; }
    // arrays
    const json& operator[](size_t index) const;            // throws json_error when out of range
    const std::vector<json>& elements() const { return arr; }
    size_t size() const;                                    // elements of an array / members of an object, else 0

    double as_double() const;
    float as_float() const { return (float)as_double(); }
    int64_t as_int() const;
    uint32_t as_u32() const;
    bool as_bool() const;
    const std::string& as_string() const;

    // compact re-serialisation (members in file order, numbers with 17 significant digits)
    std::string dump() const;

    // value of `key` if present, else the fallback
    double get_or(const std::string& key, double fallback) const { return contains(key) ? (*this)[key].as_double() : fallback; }

private:
    kind k = kind::null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<json> arr;
    std::vector<std::pair<std::string, json>> obj;  // insertion order kept; duplicate keys: the last one wins on lookup
    friend class json_parser;
};

#endif  // PTR_JSON_HPP_
