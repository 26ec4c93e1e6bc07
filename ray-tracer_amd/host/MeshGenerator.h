// Procedural stand-ins for the reference's mesh files, which are Git-LFS pointers
// with no payload offline.  Shape names match what the reference's scenes ask for.
#pragma once

#include "Scene.h"

#include <memory>
#include <string>
#include <vector>

namespace MeshGenerator
{

// `stem` is the mesh file name without extension ("CornellBox", "cube",
// "MaterialTestSpheres", "eschers_knot").  Returns an empty list for unknown names.
std::vector<std::shared_ptr<Mesh>> standInFor(const std::string& stem);

std::shared_ptr<Mesh> quadMesh(const std::string& name, const Vector& v0, const Vector& v1, const Vector& v2,
                               const Vector& v3);
std::shared_ptr<Mesh> boxMesh(const std::string& name, double halfExtent);
std::shared_ptr<Mesh> uvSphereMesh(const std::string& name, const Vector& center, double radius, int segments,
                                   int rings, bool smooth);
std::shared_ptr<Mesh> torusKnotMesh(const std::string& name, int p, int q, double majorRadius, double minorRadius,
                                    double tubeRadius, int pathSegments, int tubeSides);

}
