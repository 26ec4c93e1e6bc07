// LoadedScene -> rt_scene: resolve the object graph to world transforms and pack
// everything the photon pass reads into the plain-data description of
// include/rt_scene.h.  Object order is preserved (it decides volume tie-breaks and
// light ids, reference src/Worker.cpp:435-503,556-567).
#pragma once

#include "Scene.h"
#include "rt_scene.h"

#include <vector>

// Owns the arrays an rt_scene points into.
struct FlatScene
{
    rt_scene scene{};
    std::vector<rt_volume> volumes;
    std::vector<rt_mesh> meshes;
    std::vector<double> triPos, triNrm;
    std::vector<rt_material> materials;
    std::vector<rt_light> lights;
    std::vector<rt_camera> cameras;
    std::vector<rt_animation> animations;
    std::vector<rt_keyframe> keyframes;

    FlatScene() = default;
    FlatScene(const FlatScene&) = delete;
    FlatScene& operator=(const FlatScene&) = delete;
    void bind();  // point scene.* at the vectors
};

void flattenScene(const LoadedScene& loaded, FlatScene& out);
