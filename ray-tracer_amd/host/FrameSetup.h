// Frame-level quantities Renderer::renderFrame derives before the photon pass
// (reference src/Renderer.cpp:222-274,303-310).
#pragma once

#include "rt_scene.h"

struct FrameDerived
{
    double sceneDepthFootprint = 0.0;  // pixel footprint radius at the volumes' centroid depth
    double cellSize = 1.0;             // density-grid cell edge
    double minSplatRadius = 0.0;       // firefly floor of the splat footprint
    double keepRadius = 0.0;           // probe keep radius (= probe index cell size)
};

FrameDerived deriveFrame(const rt_scene& scene);
