#include "MeshGenerator.h"

#include <cmath>

namespace
{

// Vertex data goes through float like an OBJ file's would (tinyobjloader stores
// float attributes; reference src/ObjReader.cpp:61-67 widens them to double).
double f32(double v) { return static_cast<double>(static_cast<float>(v)); }
Vector f32(const Vector& v) { return {f32(v.x), f32(v.y), f32(v.z)}; }

Triangle makeTriangle(const Vector& a, const Vector& b, const Vector& c, const Vector& na, const Vector& nb,
                      const Vector& nc)
{
    Triangle t;
    t.a = f32(a);
    t.b = f32(b);
    t.c = f32(c);
    t.aNormal = f32(na);
    t.bNormal = f32(nb);
    t.cNormal = f32(nc);
    return t;
}

// Fan-triangulated quad (v0,v1,v2),(v0,v2,v3); the front face (the side
// cross(v1-v0, v2-v0) points to) carries the flat normal.
void addQuad(std::vector<Triangle>& out, const Vector& v0, const Vector& v1, const Vector& v2, const Vector& v3)
{
    const Vector n = Vector::cross(v1 - v0, v2 - v0).normalized();
    out.push_back(makeTriangle(v0, v1, v2, n, n, n));
    out.push_back(makeTriangle(v0, v2, v3, n, n, n));
}

}  // namespace

namespace MeshGenerator
{

std::shared_ptr<Mesh> quadMesh(const std::string& name, const Vector& v0, const Vector& v1, const Vector& v2,
                               const Vector& v3)
{
    std::vector<Triangle> t;
    addQuad(t, v0, v1, v2, v3);
    return std::make_shared<Mesh>(name, t);
}

std::shared_ptr<Mesh> boxMesh(const std::string& name, double h)
{
    std::vector<Triangle> t;
    addQuad(t, {h, -h, -h}, {h, h, -h}, {h, h, h}, {h, -h, h});      // +x
    addQuad(t, {-h, -h, h}, {-h, h, h}, {-h, h, -h}, {-h, -h, -h});  // -x
    addQuad(t, {-h, h, -h}, {-h, h, h}, {h, h, h}, {h, h, -h});      // +y
    addQuad(t, {-h, -h, h}, {-h, -h, -h}, {h, -h, -h}, {h, -h, h});  // -y
    addQuad(t, {h, -h, h}, {h, h, h}, {-h, h, h}, {-h, -h, h});      // +z
    addQuad(t, {-h, -h, -h}, {-h, h, -h}, {h, h, -h}, {h, -h, -h});  // -z
    return std::make_shared<Mesh>(name, t);
}

std::shared_ptr<Mesh> uvSphereMesh(const std::string& name, const Vector& center, double radius, int segments,
                                   int rings, bool smooth)
{
    auto point = [&](int ring, int segment) {
        const double theta = M_PI * ring / rings;          // 0 at +y pole
        const double phi = 2.0 * M_PI * segment / segments;
        return Vector{std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi)};
    };
    std::vector<Triangle> t;
    for (int r = 0; r < rings; ++r)
    {
        for (int s = 0; s < segments; ++s)
        {
            const Vector d00 = point(r, s), d01 = point(r, s + 1), d10 = point(r + 1, s), d11 = point(r + 1, s + 1);
            const Vector p00 = center + d00 * radius, p01 = center + d01 * radius, p10 = center + d10 * radius,
                         p11 = center + d11 * radius;
            // outward winding: (p00, p01, p11, p10) seen from outside is counter-clockwise
            auto emit = [&](const Vector& a, const Vector& b, const Vector& c, const Vector& na, const Vector& nb,
                            const Vector& nc) {
                const Vector g = Vector::cross(b - a, c - a);
                if (g.magnitude() < 1e-12)
                {
                    return;  // collapsed triangle at a pole
                }
                if (smooth)
                {
                    t.push_back(makeTriangle(a, b, c, na, nb, nc));
                }
                else
                {
                    const Vector n = g.normalized();
                    t.push_back(makeTriangle(a, b, c, n, n, n));
                }
            };
            emit(p00, p01, p11, d00, d01, d11);
            emit(p00, p11, p10, d00, d11, d10);
        }
    }
    return std::make_shared<Mesh>(name, t);
}

std::shared_ptr<Mesh> torusKnotMesh(const std::string& name, int p, int q, double R, double r, double tube,
                                    int pathSegments, int tubeSides)
{
    auto curve = [&](double u) {
        const double a = R + r * std::cos(q * u);
        return Vector{a * std::cos(p * u), r * std::sin(q * u), a * std::sin(p * u)};
    };
    std::vector<std::vector<Vector>> ringPoint(pathSegments), ringNormal(pathSegments);
    for (int i = 0; i < pathSegments; ++i)
    {
        const double u = 2.0 * M_PI * i / pathSegments;
        const double du = 1e-4;
        const Vector c = curve(u);
        const Vector tangent = (curve(u + du) - curve(u - du)).normalized();
        // frame from the curve's own acceleration (never parallel to the tangent on a torus knot)
        const Vector accel = curve(u + du) + curve(u - du) - c * 2.0;
        Vector normal = (accel - tangent * Vector::dot(accel, tangent)).normalized();
        const Vector binormal = Vector::cross(tangent, normal).normalized();
        for (int k = 0; k < tubeSides; ++k)
        {
            const double v = 2.0 * M_PI * k / tubeSides;
            const Vector n = normal * std::cos(v) + binormal * std::sin(v);
            ringPoint[i].push_back(c + n * tube);
            ringNormal[i].push_back(n);
        }
    }
    std::vector<Triangle> t;
    for (int i = 0; i < pathSegments; ++i)
    {
        const int i1 = (i + 1) % pathSegments;
        for (int k = 0; k < tubeSides; ++k)
        {
            const int k1 = (k + 1) % tubeSides;
            const Vector &a = ringPoint[i][k], &b = ringPoint[i1][k], &c = ringPoint[i1][k1], &d = ringPoint[i][k1];
            const Vector &na = ringNormal[i][k], &nb = ringNormal[i1][k], &nc = ringNormal[i1][k1],
                         &nd = ringNormal[i][k1];
            // choose the winding whose geometric normal agrees with the outward tube normal
            const bool flip = Vector::dot(Vector::cross(b - a, c - a), na) < 0.0;
            if (!flip)
            {
                t.push_back(makeTriangle(a, b, c, na, nb, nc));
                t.push_back(makeTriangle(a, c, d, na, nc, nd));
            }
            else
            {
                t.push_back(makeTriangle(a, c, b, na, nc, nb));
                t.push_back(makeTriangle(a, d, c, na, nd, nc));
            }
        }
    }
    return std::make_shared<Mesh>(name, t);
}

std::vector<std::shared_ptr<Mesh>> standInFor(const std::string& stem)
{
    std::vector<std::shared_ptr<Mesh>> meshes;
    if (stem == "CornellBox")
    {
        // 300-unit box, x in [-150,150], y in [0,300], z in [-150,150], open toward -z;
        // extents from the reference's CornellBoxQuads.json:61-103 (without its 5-unit
        // overshoot).  Front faces look inward.  "Right" is the x = -150 wall, as there.
        const double x = 150, y = 300, z = 150;
        meshes.push_back(quadMesh("Ceiling", {-x, y, -z}, {x, y, -z}, {x, y, z}, {-x, y, z}));
        meshes.push_back(quadMesh("Left", {x, 0, -z}, {x, 0, z}, {x, y, z}, {x, y, -z}));
        meshes.push_back(quadMesh("Right", {-x, 0, z}, {-x, 0, -z}, {-x, y, -z}, {-x, y, z}));
        meshes.push_back(quadMesh("Floor", {-x, 0, z}, {x, 0, z}, {x, 0, -z}, {-x, 0, -z}));
        meshes.push_back(quadMesh("Back", {x, 0, z}, {-x, 0, z}, {-x, y, z}, {x, y, z}));
    }
    else if (stem == "cube")
    {
        meshes.push_back(boxMesh("Cube", 50.0));
    }
    else if (stem == "MaterialTestSpheres")
    {
        // A row of test spheres, each offset inside the mesh file.  The offset and radius
        // of SphereDiffuse are read off the reference's own published render
        // renders/bombardment-20M.png (sphere of radius ~50 centred under the spot light
        // at world (-150, 5, 400) with the volume at (-60,-60,370)); flat per-face normals
        // and 32 x 16 facets match the banding visible in that image.
        meshes.push_back(uvSphereMesh("SphereDiffuse", {-90.0, 65.0, 30.0}, 50.0, 32, 16, false));
        meshes.push_back(uvSphereMesh("SphereMirror", {30.0, 65.0, 30.0}, 50.0, 32, 16, true));
        meshes.push_back(uvSphereMesh("SphereGlass", {150.0, 65.0, 30.0}, 50.0, 32, 16, true));
    }
    else if (stem == "eschers_knot")
    {
        // (2,3) torus knot tube, 384 x 20 quads = 15 360 triangles, smooth normals; about
        // the triangle count a 1.13 MB OBJ holds.
        meshes.push_back(torusKnotMesh("Knot", 2, 3, 40.0, 16.0, 7.0, 384, 20));
    }
    return meshes;
}

}  // namespace MeshGenerator
