#include "Json.h"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace Json
{

class Parser
{
public:
    explicit Parser(const std::string& text) : m_text(text) {}

    Value parseDocument()
    {
        Value v = parseValue();
        skipSpace();
        if (m_pos != m_text.size())
        {
            fail("unexpected trailing characters");
        }
        return v;
    }

private:
    [[noreturn]] void fail(const std::string& what) const
    {
        throw ParseError("JSON parse error at offset " + std::to_string(m_pos) + ": " + what);
    }

    void skipSpace()
    {
        while (m_pos < m_text.size())
        {
            const char c = m_text[m_pos];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r')
            {
                ++m_pos;
            }
            else
            {
                break;
            }
        }
    }

    char peek()
    {
        skipSpace();
        if (m_pos >= m_text.size())
        {
            fail("unexpected end of input");
        }
        return m_text[m_pos];
    }

    void expect(char c)
    {
        if (peek() != c)
        {
            fail(std::string("expected '") + c + "'");
        }
        ++m_pos;
    }

    bool consumeLiteral(const char* literal)
    {
        const size_t n = std::char_traits<char>::length(literal);
        if (m_text.compare(m_pos, n, literal) == 0)
        {
            m_pos += n;
            return true;
        }
        return false;
    }

    Value parseValue()
    {
        const char c = peek();
        Value v;
        if (c == '{')
        {
            ++m_pos;
            v.m_type = Value::Type::Object;
            if (peek() == '}')
            {
                ++m_pos;
                return v;
            }
            for (;;)
            {
                if (peek() != '"')
                {
                    fail("expected object key");
                }
                const std::string key = parseString();
                expect(':');
                v.m_object[key] = parseValue();
                const char d = peek();
                ++m_pos;
                if (d == '}')
                {
                    break;
                }
                if (d != ',')
                {
                    fail("expected ',' or '}'");
                }
            }
            return v;
        }
        if (c == '[')
        {
            ++m_pos;
            v.m_type = Value::Type::Array;
            if (peek() == ']')
            {
                ++m_pos;
                return v;
            }
            for (;;)
            {
                v.m_array.push_back(parseValue());
                const char d = peek();
                ++m_pos;
                if (d == ']')
                {
                    break;
                }
                if (d != ',')
                {
                    fail("expected ',' or ']'");
                }
            }
            return v;
        }
        if (c == '"')
        {
            v.m_type = Value::Type::String;
            v.m_string = parseString();
            return v;
        }
        if (consumeLiteral("true"))
        {
            v.m_type = Value::Type::Bool;
            v.m_bool = true;
            return v;
        }
        if (consumeLiteral("false"))
        {
            v.m_type = Value::Type::Bool;
            v.m_bool = false;
            return v;
        }
        if (consumeLiteral("null"))
        {
            return v;
        }
        if (c == '-' || (c >= '0' && c <= '9'))
        {
            return parseNumber();
        }
        fail("unexpected character");
    }

    Value parseNumber()
    {
        const size_t start = m_pos;
        bool integral = true;
        if (m_text[m_pos] == '-')
        {
            ++m_pos;
        }
        while (m_pos < m_text.size())
        {
            const char c = m_text[m_pos];
            if (c >= '0' && c <= '9')
            {
                ++m_pos;
            }
            else if (c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-')
            {
                integral = false;
                ++m_pos;
            }
            else
            {
                break;
            }
        }
        const std::string token = m_text.substr(start, m_pos - start);
        Value v;
        v.m_type = Value::Type::Number;
        char* end = nullptr;
        errno = 0;
        v.m_number = std::strtod(token.c_str(), &end);
        if (end == token.c_str() || *end != '\0')
        {
            fail("malformed number");
        }
        if (integral)
        {
            errno = 0;
            const long long i = std::strtoll(token.c_str(), &end, 10);
            if (errno == 0 && *end == '\0')
            {
                v.m_integral = true;
                v.m_integer = i;
            }
        }
        return v;
    }

    std::string parseString()
    {
        expect('"');
        std::string out;
        while (true)
        {
            if (m_pos >= m_text.size())
            {
                fail("unterminated string");
            }
            const char c = m_text[m_pos++];
            if (c == '"')
            {
                break;
            }
            if (c != '\\')
            {
                out.push_back(c);
                continue;
            }
            if (m_pos >= m_text.size())
            {
                fail("unterminated escape");
            }
            const char e = m_text[m_pos++];
            switch (e)
            {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u':
                {
                    if (m_pos + 4 > m_text.size())
                    {
                        fail("short \\u escape");
                    }
                    const unsigned code = static_cast<unsigned>(std::strtoul(m_text.substr(m_pos, 4).c_str(), nullptr, 16));
                    m_pos += 4;
                    if (code < 0x80)
                    {
                        out.push_back(static_cast<char>(code));
                    }
                    else if (code < 0x800)
                    {
                        out.push_back(static_cast<char>(0xC0 | (code >> 6)));
                        out.push_back(static_cast<char>(0x80 | (code & 0x3F)));
                    }
                    else
                    {
                        out.push_back(static_cast<char>(0xE0 | (code >> 12)));
                        out.push_back(static_cast<char>(0x80 | ((code >> 6) & 0x3F)));
                        out.push_back(static_cast<char>(0x80 | (code & 0x3F)));
                    }
                    break;
                }
                default: fail("unknown escape");
            }
        }
        return out;
    }

    const std::string& m_text;
    size_t m_pos = 0;
};

bool Value::contains(const std::string& key) const
{
    return m_type == Type::Object && m_object.count(key) > 0;
}

const Value& Value::at(const std::string& key) const
{
    if (m_type != Type::Object)
    {
        throw std::runtime_error("JSON value is not an object (looking up \"" + key + "\")");
    }
    const auto it = m_object.find(key);
    if (it == m_object.end())
    {
        throw std::runtime_error("JSON object has no key \"" + key + "\"");
    }
    return it->second;
}

const Value& Value::at(size_t index) const
{
    if (m_type != Type::Array || index >= m_array.size())
    {
        throw std::runtime_error("JSON array index out of range");
    }
    return m_array[index];
}

size_t Value::size() const
{
    if (m_type == Type::Array) return m_array.size();
    if (m_type == Type::Object) return m_object.size();
    if (m_type == Type::Null) return 0;
    return 1;
}

double Value::asDouble() const
{
    if (m_type != Type::Number)
    {
        throw std::runtime_error("JSON value is not a number");
    }
    return m_number;
}

long long Value::asInteger() const
{
    if (m_type != Type::Number)
    {
        throw std::runtime_error("JSON value is not a number");
    }
    return m_integral ? m_integer : static_cast<long long>(m_number);
}

size_t Value::asSize() const
{
    const long long i = asInteger();
    if (i < 0)
    {
        throw std::runtime_error("JSON number must be non-negative");
    }
    return static_cast<size_t>(i);
}

bool Value::asBool() const
{
    if (m_type != Type::Bool)
    {
        throw std::runtime_error("JSON value is not a boolean");
    }
    return m_bool;
}

const std::string& Value::asString() const
{
    if (m_type != Type::String)
    {
        throw std::runtime_error("JSON value is not a string");
    }
    return m_string;
}

const std::map<std::string, Value>& Value::items() const
{
    if (m_type != Type::Object)
    {
        throw std::runtime_error("JSON value is not an object");
    }
    return m_object;
}

const std::vector<Value>& Value::elements() const
{
    if (m_type != Type::Array)
    {
        throw std::runtime_error("JSON value is not an array");
    }
    return m_array;
}

Value Value::parse(const std::string& text)
{
    Parser parser(text);
    return parser.parseDocument();
}

Value Value::parseFile(const std::string& path)
{
    std::ifstream file(path);
    if (!file)
    {
        throw ParseError("cannot open " + path);
    }
    std::stringstream buffer;
    buffer << file.rdbuf();
    return parse(buffer.str());
}

}  // namespace Json
