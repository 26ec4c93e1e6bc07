// Self-contained JSON reader for scene files.
//
// The reference parses scenes with nlohmann::json, whose objects are std::map —
// iteration is KEY-SORTED, not declaration order (reference src/SceneLoader.cpp:
// 736-743,882,993-997).  That order decides volume tie-breaks, light indices and
// the primary camera, so objects here are std::map as well.
#pragma once

#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace Json
{

class ParseError : public std::runtime_error
{
public:
    using std::runtime_error::runtime_error;
};

class Value
{
public:
    enum class Type
    {
        Null,
        Bool,
        Number,
        String,
        Array,
        Object
    };

    Value() = default;

    Type type() const { return m_type; }
    bool isNull() const { return m_type == Type::Null; }
    bool isBool() const { return m_type == Type::Bool; }
    bool isNumber() const { return m_type == Type::Number; }
    bool isString() const { return m_type == Type::String; }
    bool isArray() const { return m_type == Type::Array; }
    bool isObject() const { return m_type == Type::Object; }

    bool contains(const std::string& key) const;
    const Value& at(const std::string& key) const;
    const Value& at(size_t index) const;
    size_t size() const;

    double asDouble() const;
    float asFloat() const { return static_cast<float>(asDouble()); }
    long long asInteger() const;
    size_t asSize() const;
    bool asBool() const;
    const std::string& asString() const;

    const std::map<std::string, Value>& items() const;
    const std::vector<Value>& elements() const;

    static Value parse(const std::string& text);
    static Value parseFile(const std::string& path);

private:
    friend class Parser;
    Type m_type = Type::Null;
    bool m_bool = false;
    double m_number = 0.0;
    bool m_integral = false;
    long long m_integer = 0;
    std::string m_string;
    std::vector<Value> m_array;
    std::map<std::string, Value> m_object;
};

}  // namespace Json
