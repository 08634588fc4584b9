// Navigation-only Matterport3D simulator: see mattersim_nav.hpp.  Behaviour follows the reference's
// src/lib/MatterSim.cpp (cited per function); the code is written from scratch.
#include "mattersim_nav.hpp"

#include <algorithm>
#include <cmath>
#include <ctime>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace mattersim {

namespace {

// ---- minimal JSON reader for <scan>_connectivity.json (arrays, objects, strings, numbers, bools)
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JValue> arr;
    std::vector<std::pair<std::string, JValue>> obj;
    const JValue& at(const std::string& key) const {
        for (auto& kv : obj)
            if (kv.first == key) return kv.second;
        throw std::runtime_error("MatterSim: connectivity file lacks key '" + key + "'");
    }
};

class JParser {
public:
    explicit JParser(const std::string& text) : s(text) {}
    JValue parse() {
        JValue v = value();
        ws();
        if (p != s.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string& s;
    size_t p = 0;
    [[noreturn]] void fail(const char* what) {
        throw std::runtime_error(std::string("MatterSim: bad connectivity JSON (") + what + ")");
    }
    void ws() {
        while (p < s.size() && (s[p] == ' ' || s[p] == '\n' || s[p] == '\t' || s[p] == '\r')) ++p;
    }
    bool lit(const char* w) {
        size_t n = std::char_traits<char>::length(w);
        if (s.compare(p, n, w) == 0) {
            p += n;
            return true;
        }
        return false;
    }
    JValue value() {
        ws();
        if (p >= s.size()) fail("unexpected end");
        JValue v;
        const char c = s[p];
        if (c == '[') {
            v.kind = JValue::Arr;
            ++p;
            ws();
            if (p < s.size() && s[p] == ']') { ++p; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == ']') { ++p; break; }
                fail("expected , or ]");
            }
        } else if (c == '{') {
            v.kind = JValue::Obj;
            ++p;
            ws();
            if (p < s.size() && s[p] == '}') { ++p; return v; }
            for (;;) {
                ws();
                JValue k = value();
                if (k.kind != JValue::Str) fail("object key must be a string");
                ws();
                if (p >= s.size() || s[p] != ':') fail("expected :");
                ++p;
                v.obj.emplace_back(k.str, value());
                ws();
                if (p < s.size() && s[p] == ',') { ++p; continue; }
                if (p < s.size() && s[p] == '}') { ++p; break; }
                fail("expected , or }");
            }
        } else if (c == '"') {
            v.kind = JValue::Str;
            ++p;
            while (p < s.size() && s[p] != '"') {
                if (s[p] == '\\' && p + 1 < s.size()) ++p;   // ids contain no escapes worth decoding
                v.str.push_back(s[p++]);
            }
            if (p >= s.size()) fail("unterminated string");
            ++p;
        } else if (lit("true")) {
            v.kind = JValue::Bool;
            v.b = true;
        } else if (lit("false")) {
            v.kind = JValue::Bool;
        } else if (lit("null")) {
        } else {
            size_t used = 0;
            try {
                v.num = std::stod(s.substr(p, 64), &used);
            } catch (...) {
                fail("bad number");
            }
            v.kind = JValue::Num;
            p += used;
        }
        return v;
    }
};

inline float length3(float x, float y, float z) { return std::sqrt(x * x + y * y + z * z); }
constexpr double PI = 3.14159265358979323846;

}  // namespace

Simulator::Simulator() : state(new SimState()) { generator.seed((unsigned)time(nullptr)); }

void Simulator::setCameraResolution(int w, int h) { width = w; height = h; }
void Simulator::setCameraVFOV(double v) { vfov = v; }
void Simulator::setRenderingEnabled(bool value) { if (!initialized) rendering = value; }          // MatterSim.cpp:97-101
void Simulator::setDiscretizedViewingAngles(bool value) { if (!initialized) discretizeViews = value; }  // :103-107
void Simulator::setDatasetPath(const std::string& path) { datasetPath = path; }
void Simulator::setNavGraphPath(const std::string& path) { navGraphPath = path; }

void Simulator::init() {
    if (rendering)
        throw std::runtime_error("MatterSim (navigation-only build): rendering is not available; call "
                                 "setRenderingEnabled(False) before init(), as tasks/R2R/env.py does");
    initialized = true;
}

// MatterSim.cpp:236-274: one Location per entry of <scan>_connectivity.json; position = the
// translation column of the row-major 4x4 pose (elements 3, 7, 11), kept as float.
void Simulator::loadLocationGraph() {
    if (scanLocations.count(state->scanId)) return;
    const std::string file = navGraphPath + "/" + state->scanId + "_connectivity.json";
    std::ifstream ifs(file);
    if (ifs.fail())
        throw std::invalid_argument("MatterSim: Could not open navigation graph file: " + file +
                                    ", is scan id valid?");
    std::stringstream buf;
    buf << ifs.rdbuf();
    const std::string text = buf.str();
    const JValue root = JParser(text).parse();
    std::vector<Location> locs;
    for (const JValue& vp : root.arr) {
        Location l;
        const JValue& pose = vp.at("pose");
        if (pose.arr.size() != 16) throw std::runtime_error("MatterSim: pose must have 16 entries");
        l.pos.x = (float)pose.arr[3].num;
        l.pos.y = (float)pose.arr[7].num;
        l.pos.z = (float)pose.arr[11].num;
        for (const JValue& u : vp.at("unobstructed").arr) l.unobstructed.push_back(u.b);
        l.viewpointId = vp.at("image_id").str;
        l.included = vp.at("included").b;
        locs.push_back(std::move(l));
    }
    scanLocations[state->scanId] = std::move(locs);
}

// MatterSim.cpp:276-311: candidates = unobstructed, included viewpoints inside the horizontal field
// of view; relative heading / elevation / distance w.r.t. the camera; sorted by angular distance
// from the image centre with the current location (all zeros) first.
void Simulator::populateNavigable() {
    const std::vector<Location>& locs = scanLocations[state->scanId];
    std::vector<ViewpointPtr> nav;
    nav.push_back(state->location);
    const unsigned int idx = state->location->ix;
    const double adjusted = PI / 2.0 - state->heading;
    const float cx = (float)std::cos(adjusted), cy = (float)std::sin(adjusted);   // camera horizon dir
    const double cos_half_hfov = std::cos(vfov * width / height / 2.0);
    for (unsigned int i = 0; i < locs.size(); ++i) {
        if (i == idx) continue;
        if (!(locs[idx].unobstructed[i] && locs[i].included)) continue;
        float tx = locs[i].pos.x - locs[idx].pos.x, ty = locs[i].pos.y - locs[idx].pos.y;
        const float tz = locs[i].pos.z - locs[idx].pos.z;
        const double rel_distance = length3(tx, ty, tz);
        const float lxy = length3(tx, ty, 0.f);
        const double rel_elevation = std::atan2((double)tz, (double)lxy) - state->elevation;
        const float inv = 1.0f / std::sqrt(tx * tx + ty * ty);
        const double cos_angle = (tx * inv) * cx + (ty * inv) * cy;
        if (cos_angle >= cos_half_hfov) {
            const double rel_heading = std::atan2((double)(tx * cy - ty * cx), (double)(tx * cx + ty * cy));
            auto v = std::make_shared<Viewpoint>();
            v->viewpointId = locs[i].viewpointId;
            v->ix = i;
            v->point = locs[i].pos;
            v->rel_heading = rel_heading;
            v->rel_elevation = rel_elevation;
            v->rel_distance = rel_distance;
            nav.push_back(v);
        }
    }
    std::sort(nav.begin(), nav.end(), [](const ViewpointPtr& l, const ViewpointPtr& r) {   // MatterSim.hpp:43-48
        return std::sqrt(l->rel_heading * l->rel_heading + l->rel_elevation * l->rel_elevation) <
               std::sqrt(r->rel_heading * r->rel_heading + r->rel_elevation * r->rel_elevation);
    });
    state->navigableLocations = nav;
}

// MatterSim.cpp:339-367
void Simulator::setHeadingElevation(double heading, double elevation) {
    state->heading = std::fmod(heading, PI * 2.0);
    while (state->heading < 0.0) state->heading += PI * 2.0;
    if (discretizeViews) {
        const double inc = PI * 2.0 / headingCount;
        int step = (int)std::lround(state->heading / inc);
        if (step == headingCount) step = 0;
        state->heading = (double)step * inc;
        state->elevation = elevation;
        if (state->elevation < -elevationIncrement / 2.0) {
            state->elevation = -elevationIncrement;
            state->viewIndex = step;
        } else if (state->elevation > elevationIncrement / 2.0) {
            state->elevation = elevationIncrement;
            state->viewIndex = step + 2 * headingCount;
        } else {
            state->elevation = 0.0;
            state->viewIndex = step + headingCount;
        }
    } else {
        state->elevation = std::max(std::min(elevation, maxElevation), minElevation);
    }
}

bool Simulator::setElevationLimits(double min, double max) {                  // MatterSim.cpp:369-377
    if (min < 0.0 && min > -PI / 2.0 && max > 0.0 && max < PI / 2.0) {
        minElevation = min;
        maxElevation = max;
        return true;
    }
    return false;
}

// MatterSim.cpp:379-435
void Simulator::newEpisode(const std::string& scanId, const std::string& viewpointId, double heading,
                           double elevation) {
    if (!initialized) init();
    state->step = 0;
    setHeadingElevation(heading, elevation);
    if (state->scanId != scanId) {
        state->scanId = scanId;
        loadLocationGraph();
    }
    const std::vector<Location>& locs = scanLocations[state->scanId];
    int ix = -1;
    if (viewpointId.empty()) {
        std::uniform_int_distribution<int> distribution(0, (int)locs.size() - 1);
        const int start = distribution(generator);
        ix = start;
        while (!locs[ix].included) {                    // never start at an excluded viewpoint
            if (++ix >= (int)locs.size()) ix = 0;
            if (ix == start)
                throw std::logic_error("MatterSim: ScanId: " + scanId + " has no included viewpoints!");
        }
    } else {
        for (int i = 0; i < (int)locs.size(); ++i) {
            if (locs[i].viewpointId == viewpointId) {
                if (!locs[i].included)
                    throw std::invalid_argument("MatterSim: ViewpointId: " + viewpointId +
                                                ", is excluded from the connectivity graph.");
                ix = i;
                break;
            }
        }
        if (ix < 0)
            throw std::invalid_argument("MatterSim: Could not find viewpointId: " + viewpointId +
                                        ", is viewpoint id valid?");
    }
    auto v = std::make_shared<Viewpoint>();
    v->viewpointId = locs[ix].viewpointId;
    v->ix = (unsigned int)ix;
    v->point = locs[ix].pos;
    state->location = v;
    populateNavigable();
}

// MatterSim.cpp:470-508
void Simulator::makeAction(int index, double heading, double elevation) {
    if (!initialized || index < 0 || index >= (int)state->navigableLocations.size()) {
        std::stringstream msg;
        msg << "MatterSim: Invalid action index: " << index;
        throw std::domain_error(msg.str());
    }
    state->location = state->navigableLocations[index];
    state->location->rel_heading = 0.0;
    state->location->rel_elevation = 0.0;
    state->location->rel_distance = 0.0;
    state->step += 1;
    if (discretizeViews) {                               // only the sign of the request matters
        if (heading > 0.0) heading = PI * 2.0 / headingCount;
        if (heading < 0.0) heading = -PI * 2.0 / headingCount;
        if (elevation > 0.0) elevation = elevationIncrement;
        if (elevation < 0.0) elevation = -elevationIncrement;
    }
    setHeadingElevation(state->heading + heading, state->elevation + elevation);
    populateNavigable();
}

}  // namespace mattersim
