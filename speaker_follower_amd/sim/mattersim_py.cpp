// pybind11 module `MatterSim` with the surface of the reference's src/lib_python/MatterSimPython.cpp
// (:132-164): Simulator, SimState, ViewPoint with the same attribute / method names, copy-out state
// objects, C++ exceptions mapped by pybind11's defaults.  `rgb` is None (rendering is not built).
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "mattersim_nav.hpp"

namespace py = pybind11;
using namespace mattersim;

namespace {

struct ViewPointPy {
    explicit ViewPointPy(const ViewpointPtr& p)
        : viewpointId(p->viewpointId), ix(p->ix), rel_heading(p->rel_heading),
          rel_elevation(p->rel_elevation), rel_distance(p->rel_distance) {
        point.append(p->point.x);
        point.append(p->point.y);
        point.append(p->point.z);
    }
    std::string viewpointId;
    unsigned int ix;
    py::list point;
    double rel_heading, rel_elevation, rel_distance;
};

struct SimStatePy {
    explicit SimStatePy(const SimStatePtr& s)
        : scanId(s->scanId), step(s->step), viewIndex(s->viewIndex), rgb(py::none()),
          location(s->location), heading(s->heading), elevation(s->elevation) {
        for (const auto& v : s->navigableLocations) navigableLocations.append(ViewPointPy(v));
    }
    std::string scanId;
    unsigned int step, viewIndex;
    py::object rgb;
    ViewPointPy location;
    double heading, elevation;
    py::list navigableLocations;
};

}  // namespace

PYBIND11_MODULE(MatterSim, m) {
    m.doc() = "Navigation-only Matterport3D simulator (speaker_follower_amd)";
    py::class_<ViewPointPy>(m, "ViewPoint")
        .def_readonly("viewpointId", &ViewPointPy::viewpointId)
        .def_readonly("ix", &ViewPointPy::ix)
        .def_readonly("point", &ViewPointPy::point)
        .def_readonly("rel_heading", &ViewPointPy::rel_heading)
        .def_readonly("rel_elevation", &ViewPointPy::rel_elevation)
        .def_readonly("rel_distance", &ViewPointPy::rel_distance);
    py::class_<SimStatePy>(m, "SimState")
        .def_readonly("scanId", &SimStatePy::scanId)
        .def_readonly("step", &SimStatePy::step)
        .def_readonly("rgb", &SimStatePy::rgb)
        .def_readonly("location", &SimStatePy::location)
        .def_readonly("heading", &SimStatePy::heading)
        .def_readonly("elevation", &SimStatePy::elevation)
        .def_readonly("viewIndex", &SimStatePy::viewIndex)
        .def_readonly("navigableLocations", &SimStatePy::navigableLocations);
    py::class_<Simulator>(m, "Simulator")
        .def(py::init<>())
        .def("setDatasetPath", &Simulator::setDatasetPath)
        .def("setNavGraphPath", &Simulator::setNavGraphPath)
        .def("setCameraResolution", &Simulator::setCameraResolution)
        .def("setCameraVFOV", &Simulator::setCameraVFOV)
        .def("setRenderingEnabled", &Simulator::setRenderingEnabled)
        .def("setDiscretizedViewingAngles", &Simulator::setDiscretizedViewingAngles)
        .def("init", &Simulator::init)
        .def("setSeed", &Simulator::setSeed)
        .def("setElevationLimits", &Simulator::setElevationLimits)
        .def("newEpisode", &Simulator::newEpisode, py::arg("scanId"), py::arg("viewpointId") = std::string(),
             py::arg("heading") = 0.0, py::arg("elevation") = 0.0)
        .def("getState", [](Simulator& s) { return new SimStatePy(s.getState()); },
             py::return_value_policy::take_ownership)
        .def("makeAction", &Simulator::makeAction)
        .def("close", &Simulator::close);
}
