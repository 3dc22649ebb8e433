// SceneFile.cpp — see SceneFile.h for the reference citations.
#include "SceneFile.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>

namespace MRendererHip {
namespace {

// ---- a json value tree (objects keep their members by key; duplicates: last one wins, as in nlohmann's default)
struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double number = 0.0;
    bool boolean = false;
    std::string string;
    std::vector<JsonValue> items;
    std::map<std::string, JsonValue> members;
    const JsonValue* Find(const char* key) const {
        auto it = members.find(key);
        return it == members.end() ? nullptr : &it->second;
    }
};

class JsonReader {
public:
    JsonReader(const char* text, size_t bytes) : p_(text), end_(text + bytes), begin_(text) {}
    JsonValue ParseDocument() {
        if (end_ - p_ >= 3 && (unsigned char)p_[0] == 0xEF && (unsigned char)p_[1] == 0xBB && (unsigned char)p_[2] == 0xBF) p_ += 3;   // UTF-8 BOM
        JsonValue v = ParseValue(0);
        SkipSpace();
        if (p_ != end_) Fail("trailing characters after the document");
        return v;
    }

private:
    const char *p_, *end_, *begin_;
    static constexpr int MaxDepth = 64;

    [[noreturn]] void Fail(const char* why) const {
        char buf[160];
        std::snprintf(buf, sizeof buf, "scene json: %s at byte %zu", why, (size_t)(p_ - begin_));
        throw HipException(buf);
    }
    void SkipSpace() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++;
    }
    bool Literal(const char* word) {
        const size_t n = std::strlen(word);
        if ((size_t)(end_ - p_) >= n && std::memcmp(p_, word, n) == 0) { p_ += n; return true; }
        return false;
    }
    JsonValue ParseValue(int depth) {
        if (depth > MaxDepth) Fail("nesting too deep");
        SkipSpace();
        if (p_ == end_) Fail("unexpected end of input");
        JsonValue v;
        const char c = *p_;
        if (c == '{') {
            v.kind = JsonValue::Object;
            p_++;
            SkipSpace();
            if (p_ < end_ && *p_ == '}') { p_++; return v; }
            for (;;) {
                SkipSpace();
                if (p_ == end_ || *p_ != '"') Fail("expected a member name");
                std::string key = ParseString();
                SkipSpace();
                if (p_ == end_ || *p_ != ':') Fail("expected ':'");
                p_++;
                v.members[key] = ParseValue(depth + 1);
                SkipSpace();
                if (p_ == end_) Fail("unterminated object");
                if (*p_ == ',') { p_++; continue; }
                if (*p_ == '}') { p_++; return v; }
                Fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.kind = JsonValue::Array;
            p_++;
            SkipSpace();
            if (p_ < end_ && *p_ == ']') { p_++; return v; }
            for (;;) {
                v.items.push_back(ParseValue(depth + 1));
                SkipSpace();
                if (p_ == end_) Fail("unterminated array");
                if (*p_ == ',') { p_++; continue; }
                if (*p_ == ']') { p_++; return v; }
                Fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JsonValue::String; v.string = ParseString(); return v; }
        if (Literal("true")) { v.kind = JsonValue::Bool; v.boolean = true; return v; }
        if (Literal("false")) { v.kind = JsonValue::Bool; return v; }
        if (Literal("null")) return v;
        if (c == '-' || (c >= '0' && c <= '9')) { v.kind = JsonValue::Number; v.number = ParseNumber(); return v; }
        Fail("unexpected character");
    }
    double ParseNumber() {   // RFC 8259 grammar checked by hand, value by strtod on a bounded copy
        const char* s = p_;
        if (p_ < end_ && *p_ == '-') p_++;
        if (p_ == end_ || *p_ < '0' || *p_ > '9') Fail("malformed number");
        if (*p_ == '0') p_++; else while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
        if (p_ < end_ && *p_ == '.') {
            p_++;
            if (p_ == end_ || *p_ < '0' || *p_ > '9') Fail("malformed fraction");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
        }
        if (p_ < end_ && (*p_ == 'e' || *p_ == 'E')) {
            p_++;
            if (p_ < end_ && (*p_ == '+' || *p_ == '-')) p_++;
            if (p_ == end_ || *p_ < '0' || *p_ > '9') Fail("malformed exponent");
            while (p_ < end_ && *p_ >= '0' && *p_ <= '9') p_++;
        }
        const std::string copy(s, p_);
        return std::strtod(copy.c_str(), nullptr);
    }
    static void AppendUtf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
        else if (cp < 0x10000) { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
        else { out += (char)(0xF0 | (cp >> 18)); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    }
    unsigned Hex4() {
        if (end_ - p_ < 4) Fail("truncated \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; i++, p_++) {
            const char c = *p_;
            v = v * 16 + (c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : (Fail("bad hex digit"), 0));
        }
        return v;
    }
    std::string ParseString() {
        std::string out;
        p_++;   // opening quote
        for (;;) {
            if (p_ == end_) Fail("unterminated string");
            const unsigned char c = (unsigned char)*p_++;
            if (c == '"') return out;
            if (c < 0x20) { p_--; Fail("control character in a string"); }
            if (c != '\\') { out += (char)c; continue; }
            if (p_ == end_) Fail("unterminated escape");
            const char e = *p_++;
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
                    unsigned cp = Hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF) {   // surrogate pair
                        if (end_ - p_ < 2 || p_[0] != '\\' || p_[1] != 'u') Fail("lone surrogate");
                        p_ += 2;
                        const unsigned lo = Hex4();
                        if (lo < 0xDC00 || lo > 0xDFFF) Fail("bad low surrogate");
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    AppendUtf8(out, cp);
                    break;
                }
                default: p_--; Fail("unknown escape");
            }
        }
    }
};

[[noreturn]] void Shape(size_t index, const char* why) {
    char buf[160];
    std::snprintf(buf, sizeof buf, "scene json: mSceneLight[%zu]: %s", index, why);
    throw HipException(buf);
}

float Float(const JsonValue* v, size_t index, const char* what) {
    if (!v || v->kind != JsonValue::Number) Shape(index, what);
    return (float)v->number;   // the reference's members are float: json_value.get<float>()
}

Vector3 Vec3(const JsonValue* v, size_t index, const char* what) {
    if (!v || v->kind != JsonValue::Object) Shape(index, what);
    return Vector3{Float(v->Find("x"), index, what), Float(v->Find("y"), index, what), Float(v->Find("z"), index, what)};
}

}  // namespace

std::vector<SceneLightRecord> ParseSceneLights(const char* text, size_t bytes) {
    const JsonValue doc = JsonReader(text, bytes).ParseDocument();
    if (doc.kind != JsonValue::Object) throw HipException("scene json: the document is not an object");
    std::vector<SceneLightRecord> out;
    const JsonValue* lights = doc.Find("mSceneLight");
    if (!lights) return out;                                                  // a scene without the member has no lights
    if (lights->kind != JsonValue::Array) throw HipException("scene json: mSceneLight is not an array");
    for (size_t i = 0; i < lights->items.size(); i++) {
        const JsonValue& l = lights->items[i];
        if (l.kind != JsonValue::Object) Shape(i, "not an object");
        SceneLightRecord rec;
        const JsonValue* base = l.Find("@SceneObject");                        // Serialization.h: base class under "@<Base>"
        if (!base || base->kind != JsonValue::Object) Shape(i, "no \"@SceneObject\" member");
        if (const JsonValue* n = base->Find("mName")) {
            if (n->kind != JsonValue::String) Shape(i, "mName is not a string");
            rec.Name = n->string;
        }
        rec.Translation = Vec3(base->Find("mTranslation"), i, "mTranslation is not {x, y, z}");
        rec.Rotation = Vec3(base->Find("mRotation"), i, "mRotation is not {x, y, z}");
        rec.Scale = Vec3(base->Find("mScale"), i, "mScale is not {x, y, z}");
        rec.Color = Vec3(l.Find("mColor"), i, "mColor is not {x, y, z}");
        rec.Radius = Float(l.Find("mRadius"), i, "mRadius is not a number");
        rec.Intensity = Float(l.Find("mIntensity"), i, "mIntensity is not a number");
        if (!(rec.Radius > 0.0f) || !(rec.Intensity >= 0.0f)) Shape(i, "radius must be > 0 and intensity >= 0");
        out.push_back(rec);
    }
    return out;
}

std::vector<SceneLightRecord> LoadSceneLights(const std::string& path) {
    std::unique_ptr<FILE, int (*)(FILE*)> f(std::fopen(path.c_str(), "rb"), &std::fclose);
    if (!f) throw HipException("scene json: cannot open " + path);
    std::string text;
    char buf[1 << 16];
    for (size_t n; (n = std::fread(buf, 1, sizeof buf, f.get())) > 0;) {
        text.append(buf, n);
        if (text.size() > ((size_t)256 << 20)) throw HipException("scene json: file larger than 256 MiB");
    }
    return ParseSceneLights(text.data(), text.size());
}

void AddSceneLights(Scene* scene, const std::vector<SceneLightRecord>& records) {
    scene->ClearLights();
    for (const SceneLightRecord& r : records) {
        SceneLight light(r.Translation, r.Color, r.Radius, r.Intensity);
        // SceneObject::PostDeserialized (Scene.cpp:31-36): model matrix = FromEulerAngle(rotation in degrees) * scale + translation;
        // GetWorldBound (Scene.h:31) = matrix * local bound = min / max of the two transformed CORNERS (MathLib.cpp:5-10)
        if (r.Rotation.x != 0.0f || r.Rotation.y != 0.0f || r.Rotation.z != 0.0f || r.Scale.x != 1.0f || r.Scale.y != 1.0f || r.Scale.z != 1.0f) {
            constexpr float Deg2Rad = 3.14159265359f / 180.0f;
            const float ca = std::cos(r.Rotation.x * Deg2Rad), sa = std::sin(r.Rotation.x * Deg2Rad);
            const float cb = std::cos(r.Rotation.y * Deg2Rad), sb = std::sin(r.Rotation.y * Deg2Rad);
            const float cc = std::cos(r.Rotation.z * Deg2Rad), sc = std::sin(r.Rotation.z * Deg2Rad);
            const float rot[3][3] = {{ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc},
                                     {sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc},
                                     {-sb, cb * sc, cb * cc}};
            const float s[3] = {r.Scale.x, r.Scale.y, r.Scale.z};
            const float t[3] = {r.Translation.x, r.Translation.y, r.Translation.z};
            const float cr = r.Radius * SceneLight::CullingRadiusCoefficient * std::sqrt(r.Intensity);
            float lo[3], hi[3];
            for (int row = 0; row < 3; row++) {
                float a = t[row], b = t[row];
                for (int col = 0; col < 3; col++) { a += rot[row][col] * s[col] * -cr; b += rot[row][col] * s[col] * cr; }
                lo[row] = std::fmin(a, b);
                hi[row] = std::fmax(a, b);
            }
            light.SetWorldBound(AABB{{lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}});
        }
        scene->AddLight(light);
    }
}

}  // namespace MRendererHip
