// Navigation-only Matterport3D simulator.
//
// A from-scratch C++17 implementation of the graph-stepping half of the reference's simulator --
// the mode the R2R training code uses (rendering disabled, tasks/R2R/env.py:243) -- with the same
// public API and state semantics as include/MatterSim.hpp:101-239 / src/lib/MatterSim.cpp of the
// reference, but without OpenCV, OpenGL/OSMesa, GLM or jsoncpp: positions are plain float triples,
// the connectivity files are read by the small JSON reader in mattersim_nav.cpp.  Rendering is not
// built: setRenderingEnabled(true) followed by init() throws.
#pragma once
#include <map>
#include <memory>
#include <random>
#include <string>
#include <vector>

namespace mattersim {

struct Point3f {
    float x = 0, y = 0, z = 0;
};

struct Viewpoint {                 // MatterSim.hpp:27-40
    std::string viewpointId;
    unsigned int ix = 0;           // index into the connectivity graph
    Point3f point;                 // world coordinates
    double rel_heading = 0;        // relative to the camera
    double rel_elevation = 0;
    double rel_distance = 0;
};
typedef std::shared_ptr<Viewpoint> ViewpointPtr;

struct SimState {                  // MatterSim.hpp:53-75 (no rgb / depth: rendering is not built)
    std::string scanId;
    unsigned int step = 0;
    ViewpointPtr location;
    double heading = 0;
    double elevation = 0;
    unsigned int viewIndex = 0;    // [0-11] down, [12-23] horizon, [24-35] up (discretized views only)
    std::vector<ViewpointPtr> navigableLocations;   // [0] = stay; rest sorted by angular distance
};
typedef std::shared_ptr<SimState> SimStatePtr;

struct Location {                  // MatterSim.hpp:81-93
    bool included = false;
    std::string viewpointId;
    Point3f pos;
    std::vector<bool> unobstructed;
};

class Simulator {
public:
    Simulator();
    void setCameraResolution(int width, int height);
    void setCameraVFOV(double vfov);
    void setRenderingEnabled(bool value);
    void setDiscretizedViewingAngles(bool value);
    void init();
    void setDatasetPath(const std::string& path);
    void setNavGraphPath(const std::string& path);
    void setSeed(int seed) { generator.seed(seed); }
    bool setElevationLimits(double min, double max);
    void newEpisode(const std::string& scanId, const std::string& viewpointId = std::string(),
                    double heading = 0, double elevation = 0);
    SimStatePtr getState() { return state; }
    void makeAction(int index, double heading, double elevation);
    void close() { initialized = false; }
    bool renderingEnabled() const { return rendering; }
    int imageWidth() const { return width; }
    int imageHeight() const { return height; }

private:
    static constexpr int headingCount = 12;                                   // MatterSim.hpp:185
    static constexpr double elevationIncrement = 3.14159265358979323846 / 6;  // :186
    void loadLocationGraph();
    void populateNavigable();
    void setHeadingElevation(double heading, double elevation);

    SimStatePtr state;
    bool initialized = false, rendering = true, discretizeViews = false;
    int width = 320, height = 240;
    double vfov = 0.8, minElevation = -0.94, maxElevation = 0.94;
    std::string datasetPath = "./data", navGraphPath = "./connectivity";
    std::map<std::string, std::vector<Location>> scanLocations;
    std::default_random_engine generator;
};

}  // namespace mattersim
