// rt_device.h — device-side geometry, BSDF and emission routines (gfx950).
//
// Every routine reproduces, operation for operation, the arithmetic of the
// reference function it cites (paths relative to the reference checkout), in FP64
// for geometry and FP32 for energy, so that a photon's path is the same bits the
// reference's CPU code would produce from the same inputs.  That requires:
//   * -ffp-contract=off  (no FMA contraction anywhere; the watertight triangle
//     test depends on individually rounded products, src/Ray.cpp:87-98);
//   * IEEE division and square root (hipcc's default for f64);
//   * the shared sin/cos/acos and Philox stream of include/rt_portable.h.
// Vector semantics follow the reference's AVX path: dot = (x*x + y*y) + z*z, unary
// minus = 0 - v, normalise = lane-wise divide by sqrt(dot) (src/Vector.cpp:233-267,
// 340-347).
#pragma once

#include <hip/hip_runtime.h>

#include "rt_portable.h"
#include "rt_scene.h"

#define RT_DEV __device__ __forceinline__

// Optional divergence census (build with -DRT_DIAG): wave-level trip counts and active
// lanes of the traversal loops, printed when a tracer is destroyed.
#ifdef RT_DIAG
__device__ unsigned long long g_rtDiag[32];
#endif
#if defined(RT_DIAG) && RT_DIAG == 1
#define RT_DIAG_TRIP(slot)                                                                     \
    do                                                                                         \
    {                                                                                          \
        const unsigned long long diagMask = __ballot(1);                                       \
        if (static_cast<int>(threadIdx.x & 63) == __builtin_ctzll(diagMask))                   \
        {                                                                                      \
            atomicAdd(&g_rtDiag[slot], 1ull);                                                  \
            atomicAdd(&g_rtDiag[(slot) + 1], static_cast<unsigned long long>(__builtin_popcountll(diagMask))); \
        }                                                                                      \
    } while (0)
#else
#define RT_DIAG_TRIP(slot) do { } while (0)
#endif
// Cycle census (build with -DRT_DIAG=2): wave cycles between RT_DIAG_TIC(name) and
// RT_DIAG_TOC(name, slot), summed per wave in LDS while the kernel runs and added to
// g_rtDiag[slot] at its end (per-trip global atomics would dominate what they measure).
#if defined(RT_DIAG) && RT_DIAG == 2
__device__ __forceinline__ unsigned long long* rtDiagWaveSlots()
{
    __shared__ unsigned long long rtDiagCycles[4 * 16];
    return rtDiagCycles + 16 * (threadIdx.x >> 6);
}
#define RT_DIAG_TIC(name) const long long name = __builtin_readcyclecounter()
#define RT_DIAG_TOC(name, slot)                                                                 \
    do                                                                                          \
    {                                                                                           \
        const long long diagCycles = __builtin_readcyclecounter() - name;                      \
        if (static_cast<int>(threadIdx.x & 63) == __builtin_ctzll(__ballot(1)))                \
            rtDiagWaveSlots()[slot] += static_cast<unsigned long long>(diagCycles);            \
    } while (0)
#define RT_DIAG_WAVE_BEGIN                                                                      \
    do                                                                                          \
    {                                                                                           \
        if ((threadIdx.x & 63) < 16) rtDiagWaveSlots()[threadIdx.x & 63] = 0;                  \
    } while (0)
#define RT_DIAG_WAVE_END                                                                        \
    do                                                                                          \
    {                                                                                           \
        if ((threadIdx.x & 63) < 16) atomicAdd(&g_rtDiag[threadIdx.x & 63], rtDiagWaveSlots()[threadIdx.x & 63]); \
    } while (0)
#else
#define RT_DIAG_TIC(name) do { } while (0)
#define RT_DIAG_TOC(name, slot) do { } while (0)
#define RT_DIAG_WAVE_BEGIN do { } while (0)
#define RT_DIAG_WAVE_END do { } while (0)
#endif

namespace rtd
{

constexpr double kPi = 3.14159265358979323846;
constexpr double kPi2 = 3.14159265358979323846 * 2.0;
constexpr double kEps = 2.220446049250313e-16;  // std::numeric_limits<double>::epsilon()
constexpr double kMaxDouble = 1.7976931348623157e308;

struct Vec
{
    double x, y, z;
};

RT_DEV Vec mk(double x, double y, double z) { return Vec{x, y, z}; }
RT_DEV Vec operator+(const Vec& a, const Vec& b) { return Vec{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_DEV Vec operator-(const Vec& a, const Vec& b) { return Vec{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_DEV Vec neg(const Vec& v) { return Vec{0.0 - v.x, 0.0 - v.y, 0.0 - v.z}; }
RT_DEV Vec operator*(const Vec& a, double s) { return Vec{a.x * s, a.y * s, a.z * s}; }
RT_DEV Vec operator*(double s, const Vec& a) { return Vec{s * a.x, s * a.y, s * a.z}; }
RT_DEV Vec operator/(const Vec& a, double s) { return Vec{a.x / s, a.y / s, a.z / s}; }
RT_DEV double dot(const Vec& a, const Vec& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV double magnitude(const Vec& a) { return __builtin_sqrt(dot(a, a)); }
RT_DEV Vec normalized(const Vec& a)
{
    const double root = __builtin_sqrt(dot(a, a));
    return Vec{a.x / root, a.y / root, a.z / root};
}
RT_DEV Vec cross(const Vec& a, const Vec& b)
{
    return Vec{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
RT_DEV Vec reflected(const Vec& i, const Vec& n)  // src/Vector.cpp:298-312
{
    const double dot2 = 2 * dot(i, n);
    return Vec{i.x - dot2 * n.x, i.y - dot2 * n.y, i.z - dot2 * n.z};
}
RT_DEV double axisOf(const Vec& v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }
RT_DEV double dmin(double a, double b) { return b < a ? b : a; }  // std::min
RT_DEV double dmax(double a, double b) { return a < b ? b : a; }  // std::max
RT_DEV double dabs(double a) { return __builtin_fabs(a); }

struct Quat
{
    double x, y, z, w;
};

// src/Quaternion.cpp:88-109
RT_DEV Quat qmul(const Quat& l, const Quat& r)
{
    return Quat{l.w * r.x + l.x * r.w + l.y * r.z - l.z * r.y, l.w * r.y + l.y * r.w + l.z * r.x - l.x * r.z,
                l.w * r.z + l.z * r.w + l.x * r.y - l.y * r.x, l.w * r.w - l.x * r.x - l.y * r.y - l.z * r.z};
}
RT_DEV Vec rotate(const Quat& q, const Vec& v)
{
    const Quat p{v.x, v.y, v.z, 0.0};
    const Quat c{-q.x, -q.y, -q.z, q.w};
    const Quat r = qmul(qmul(q, p), c);
    return Vec{r.x, r.y, r.z};
}

struct Color
{
    float r, g, b;
};
RT_DEV Color operator*(const Color& a, float s) { return Color{a.r * s, a.g * s, a.b * s}; }
RT_DEV Color operator*(const Color& a, const Color& b) { return Color{a.r * b.r, a.g * b.g, a.b * b.b}; }
RT_DEV float brightness(const Color& c) { return c.r + c.g + c.b; }

// ------------------------------------------------------------------ flattened scene
// One inner node of a mesh's tree (leaves are not stored): both children's boxes in single
// precision, rounded outward, and one reference per child — the index of an inner node when
// >= 0, kNoChild, or a leaf's triangle slots packed as -1 - (first * 16 + count - 1).  The
// boxes only decide which triangles are reached; which hit is reported is decided per
// triangle (SceneView::triRec), so they need to contain, not to match the reference's.
struct Node
{
    float leftBox[6];  // min xyz, max xyz
    float rightBox[6];
    int32_t left, right;
    int32_t reserved[2];
};
static_assert(sizeof(Node) == 64, "Node must be one 64-byte line");
constexpr int32_t kNoChild = static_cast<int32_t>(0x80000000u);

// Volume::type beyond the scene description's RT_VOLUME_*: the first of several flat meshes of one pose,
// which is cast for all of them through a tree over all their triangles (SceneView::triOwner tells
// whose triangle was hit), and the others of them, which the loop over the volumes passes over.
constexpr int32_t kVolumeMeshGroup = 16, kVolumePassedOver = 17;

struct Volume
{
    int32_t type, material, nodeBase, nodeCount;
    int64_t triBase;  // first triangle of the mesh in the page-ordered arrays
    int32_t animation;       // index into SceneView::animations, or -1 for a static volume
    int32_t hasWorldBox;     // worldBox is set (static mesh, sphere or quad)
    // Box around the volume in WORLD space, grown far beyond the rounding of the transforms: a ray
    // that misses it, or enters it beyond the nearest hit so far, cannot hit the volume, and
    // the cast skips it without transforming the ray.
    float worldBox[6];
    double extent;           // largest |coordinate| of the mesh (scale of the pruning slack)
    // What the trace kernel keeps of this mesh in LDS (SceneView::cacheImage): its first
    // ldsNodeCount nodes (breadth-first numbering: the top of the tree) at byte ldsNodeOffset of the
    // image, and, when ldsTriOffset >= 0, all its triangles as 80-byte records (vertices + word).
    int32_t ldsNodeCount, ldsNodeOffset, ldsTriOffset;
    // Volumes with the same number share position and rotation bit for bit (and are not keyframed):
    // a ray taken into the object space of one of them is in the object space of all of them.
    int32_t poseId;
    double pos[3];
    Quat rot, inv;    // rotation and Quaternion::inverse() of it (src/Quaternion.cpp:49-52)
    double param[18];
};

struct Material
{
    int32_t type, pad;
    float color[3];
    float pad2;
    double alpha, ior;
};

struct Light
{
    int32_t type, shape;
    double pos[3];
    Quat rot;
    double forward[3];  // rot * unitZ
    float color[3];
    float pad;
    double flux, innerRadius, maxAngle, radius, width, height;
};

struct Camera
{
    double pos[3];
    double right[3], up[3], forward[3];
    double aspect, halfHeight, tanPixelHalfAngle;
    double orthographicHeight, apertureRadius, focusDistance;
    int32_t width, height, bounceFilter, lightFilter, projection;
    int32_t animation;  // index into SceneView::animations, or -1 for a static camera
    float windowStart, windowEnd;
    float* buffer;  // [h][w][3]
};

// A visible area-light face (reference include/EmitterPatch.h, src/EmitterPatch.cpp:15-53).
struct EmitterPatch
{
    double center[3], right[3], up[3], normal[3];
    double halfWidth, halfHeight, radius;
    float radiance[3];
    int32_t isDisc;
};

// One camera sample that reached a non-delta surface (reference include/ProbeGather.h:66-79).
struct GatherPoint
{
    int32_t px, py;
    double position[3], normal[3], viewDir[3];
    double unfoldedPathLength, footprintRadius;
    float throughput[3];
    int32_t material;  // -1: emitter face
    float sampleTime, sampleWeight;
};

struct SceneView
{
    const Volume* volumes;
    const Node* nodes;
    // [16 * T], one 128-byte record per triangle slot: vertices a b c (9), a word with the
    // triangle's rank in the reference's visiting order (low half) and the barycentric weight
    // from which a hit is certainly admitted by the reference's box (float, high half), then that
    // box: min xyz, max xyz of the reference node whose page holds the triangle (6)
    const double* triRec;
    const double* triNrm;  // [9 * T] page order
    const int32_t* triId;  // slot -> index within the mesh (load order)
    const int32_t* triOwner;  // slot -> the volume the triangle belongs to, for the trees of kVolumeMeshGroup (else -1)
    const Material* materials;
    const Light* lights;
    Camera* cameras;
    const EmitterPatch* patches;
    int32_t patchCount;
    int32_t stackDepth;  // traversal stack entries per lane (inner levels of the deepest tree)
    int32_t volumeCount, lightCount, cameraCount, bounceThreshold;
    int32_t materialCount, meshVolumeCount;
    int32_t groupCount;  // volumes of kVolumeMeshGroup
    double terminationThreshold;
    double splatLuminanceClamp;
    uint64_t photonsPerLight;
    uint32_t seed;
    float windowStart, windowEnd;  // primary camera's exposure window
    int32_t hasCamera;
    // probe pass
    float frameTime, shutterSpan;
    int32_t cameraTimeSamples;
    uint32_t probeSeed;
    // keyframed poses (reference include/AnimationQuery.h:94-173); animationCount == 0: static scene
    const rt_animation* animations;
    const rt_keyframe* keyframes;
    int32_t animationCount;
    // The part of the scene the trace kernel copies into LDS behind its traversal stacks: the top of
    // every mesh's tree and the triangles of the meshes that fit (Volume::lds*).  cacheBytes is a
    // multiple of 16; 0 = nothing is cached.
    uint32_t cacheBytes;
    const unsigned char* cacheImage;
};

struct Ray
{
    Vec o, d;
};

struct Hit
{
    Vec position, normal;
    double distance;
    int32_t material, volume, triangle;
};

// ------------------------------------------------------------------ intersectors
// src/Ray.cpp:16-57.  Returns the entry parameter through *entry (>= 0).
RT_DEV bool rayIntersectsBounds(const Ray& ray, const double* bmin, const double* bmax, double* entry)
{
    double tmin = 0.0;
    double tmax = kMaxDouble;
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        const double origin = axisOf(ray.o, i);
        const double direction = axisOf(ray.d, i);
        if (dabs(direction) < kEps)
        {
            if (origin < bmin[i] || origin > bmax[i])
            {
                return false;
            }
        }
        else
        {
            const double ood = 1.0 / direction;
            double t1 = (bmin[i] - origin) * ood;
            double t2 = (bmax[i] - origin) * ood;
            if (t1 > t2)
            {
                const double t = t1;
                t1 = t2;
                t2 = t;
            }
            tmin = dmax(tmin, t1);
            tmax = dmin(tmax, t2);
            if (tmin > tmax)
            {
                return false;
            }
        }
    }
    *entry = tmin;
    return true;
}

// Per-ray constants of the slab test (they depend on the ray only).  A box is six doubles (min xyz,
// max xyz); nearIdx[i] / farIdx[i] pick the plane of axis i the ray meets first / last, which is what
// the reference's "swap when t1 > t2" amounts to: the order of the two products follows the sign of
// the reciprocal.  ood is 1 / direction, or +infinity on an axis the reference treats as parallel
// (|direction| < epsilon): there (plane - origin) * inf is -inf / +inf while the origin lies between
// the planes (or NaN on a plane, which the IEEE max / min below ignore) and drives the interval empty
// when it lies outside, exactly the reference's containment test (src/Ray.cpp:24-32).
struct SlabPrep
{
    double ood[3];
    int32_t nearIdx[3], farIdx[3];
};

RT_DEV SlabPrep prepareSlab(const Ray& ray)
{
    SlabPrep p;
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        const double d = axisOf(ray.d, i);
        const bool parallel = dabs(d) < kEps;
        const double ood = 1.0 / d;
        p.ood[i] = parallel ? __builtin_inf() : ood;
        const bool reversed = !parallel && ood < 0.0;
        p.nearIdx[i] = reversed ? i + 3 : i;
        p.farIdx[i] = reversed ? i : i + 3;
    }
    return p;
}

// Per-ray constants of the watertight test.  (The three reciprocals of the slab test are not among
// them: a cast needs the double-precision slab test for few of its rays, see castMesh.)
struct RayPrep
{
    int kx, ky, kz;      // watertight axis permutation, src/Ray.cpp:100-139
    double sx, sy, sz;   // shear constants, :141-147
    double okx, oky, okz;  // ray origin in permuted order
    bool parallel;         // some axis is treated as parallel by the slab test
};

RT_DEV RayPrep prepareRay(const Ray& ray)
{
    RayPrep p;
    p.parallel = dabs(ray.d.x) < kEps || dabs(ray.d.y) < kEps || dabs(ray.d.z) < kEps;
    const double adx = dabs(ray.d.x), ady = dabs(ray.d.y), adz = dabs(ray.d.z);
    int kz = 2;
    if (adx > ady && adx > adz)
    {
        kz = 0;
    }
    else if (ady > adz)
    {
        kz = 1;
    }
    int kx = kz + 1;
    if (kx == 3) kx = 0;
    int ky = kx + 1;
    if (ky == 3) ky = 0;
    if (axisOf(ray.d, kz) < 0.0)
    {
        const int t = kx;
        kx = ky;
        ky = t;
    }
    p.kx = kx;
    p.ky = ky;
    p.kz = kz;
    const double dkx = axisOf(ray.d, kx), dky = axisOf(ray.d, ky), dkz = axisOf(ray.d, kz);
    p.sx = dkx / dkz;
    p.sy = dky / dkz;
    p.sz = 1.0 / dkz;
    p.okx = axisOf(ray.o, kx);
    p.oky = axisOf(ray.o, ky);
    p.okz = axisOf(ray.o, kz);
    return p;
}

// Slab test with the per-ray constants hoisted; same decision and same entry parameter
// as rayIntersectsBounds.  The reference leaves its loop at the first axis whose
// interval is empty; the interval only ever shrinks, so testing it once after the last
// axis gives the same answer.  `box` points at six doubles in memory (min xyz, max xyz).
RT_DEV bool slabTest(const Ray& ray, const SlabPrep& p, const double* box, double* entry)
{
    double tmin = 0.0;
    double tmax = kMaxDouble;
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        const double origin = axisOf(ray.o, i);
        const double tnear = (box[p.nearIdx[i]] - origin) * p.ood[i];
        const double tfar = (box[p.farIdx[i]] - origin) * p.ood[i];
        tmin = __builtin_fmax(tmin, tnear);
        tmax = __builtin_fmin(tmax, tfar);
    }
    *entry = tmin;
    return !(tmin > tmax);
}

RT_DEV bool edgeAccept(double value, double ex, double ey)  // src/Ray.cpp:190-202
{
    if (value > 0.0) return true;
    if (value < 0.0) return false;
    return (ey < 0.0) || (ey == 0.0 && ex < 0.0);
}

// Watertight ray/triangle test, src/Ray.cpp:85-247.  On a hit writes the barycentric
// weights, the reconstructed position and the Euclidean distance.  `tp` points at the
// nine vertex coordinates in memory; the sheared coordinates need the vertices in the
// ray's permuted axis order, which is read straight from memory by index (selecting among
// registers after wide loads of the record was measured 10 % slower for the whole kernel,
// and a scalar-cache path for waves that share triangle and permutation 3 % slower).
RT_DEV bool rayIntersectsTriangle(const Ray& ray, const RayPrep& p, const double* tp, double* bary, Vec* position,
                                  double* distance)
{
    const double aKz = tp[p.kz] - p.okz, bKz = tp[3 + p.kz] - p.okz, cKz = tp[6 + p.kz] - p.okz;
    const double ax = (tp[p.kx] - p.okx) - p.sx * aKz, ay = (tp[p.ky] - p.oky) - p.sy * aKz;
    const double bx = (tp[3 + p.kx] - p.okx) - p.sx * bKz, by = (tp[3 + p.ky] - p.oky) - p.sy * bKz;
    const double cx = (tp[6 + p.kx] - p.okx) - p.sx * cKz, cy = (tp[6 + p.ky] - p.oky) - p.sy * cKz;

    const double u = cx * by - cy * bx;
    const double v = ax * cy - ay * cx;
    const double w = bx * ay - by * ax;

    if (!edgeAccept(w, bx - ax, by - ay) || !edgeAccept(u, cx - bx, cy - by) || !edgeAccept(v, ax - cx, ay - cy))
    {
        return false;
    }
    const double det = u + v + w;
    if (det <= 0.0)
    {
        return false;
    }
    const double az = p.sz * aKz, bz = p.sz * bKz, cz = p.sz * cKz;
    const double tScaled = u * az + v * bz + w * cz;
    if (tScaled <= 0.0)
    {
        return false;
    }
    const Vec a = mk(tp[0], tp[1], tp[2]), b = mk(tp[3], tp[4], tp[5]), c = mk(tp[6], tp[7], tp[8]);
    const double invDet = 1.0 / det;
    const double ca = u * invDet, cb = v * invDet, cc = w * invDet;
    const Vec pos = (ca * a) + (cb * b) + (cc * c);  // src/Triangle.cpp:42-45
    bary[0] = ca;
    bary[1] = cb;
    bary[2] = cc;
    *position = pos;
    *distance = magnitude(pos - ray.o);
    return true;
}

RT_DEV Vec triangleNormal(const double* tn, const double* bary)  // src/Triangle.cpp:47-50, src/Ray.cpp:243
{
    const Vec aN = mk(tn[0], tn[1], tn[2]), bN = mk(tn[3], tn[4], tn[5]), cN = mk(tn[6], tn[7], tn[8]);
    return normalized((bary[0] * aN) + (bary[1] * bN) + (bary[2] * cN));
}

// src/Ray.cpp:249-271
RT_DEV bool rayIntersectsPlane(const Ray& ray, const Vec& normal, double planeDot, Vec* position, Vec* outNormal,
                               double* distance)
{
    const double d = dot(normal, ray.d);
    if (dabs(d) <= kEps)
    {
        return false;
    }
    const double t = (planeDot - dot(normal, ray.o)) / d;
    if (t < 0.0)
    {
        return false;
    }
    *position = ray.o + (t * ray.d);
    *outNormal = normal;
    *distance = magnitude(*position - ray.o);
    return true;
}

// src/Ray.cpp:273-327
RT_DEV bool rayIntersectsSphere(const Ray& ray, const Vec& center, double radius, Vec* position, Vec* normal,
                                double* distance)
{
    const Vec oc = ray.o - center;
    const double a = dot(ray.d, ray.d);
    if (dabs(a) <= kEps)
    {
        return false;
    }
    const double b = 2.0 * dot(oc, ray.d);
    const double c = dot(oc, oc) - radius * radius;
    const double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0)
    {
        return false;
    }
    const double sqrtDiscriminant = __builtin_sqrt(discriminant);
    const double inv = 1.0 / (2.0 * a);
    double t = (-b - sqrtDiscriminant) * inv;
    if (t <= 1e-6)
    {
        t = (-b + sqrtDiscriminant) * inv;
        if (t <= 1e-6)
        {
            return false;
        }
    }
    *position = ray.o + (t * ray.d);
    *normal = normalized(*position - center);
    *distance = magnitude(*position - ray.o);
    return true;
}

// src/Ray.cpp:329-375; q = origin edgeU edgeV normal invU invV
RT_DEV bool rayIntersectsQuad(const Ray& ray, const double* q, Vec* position, Vec* normal, double* distance, Vec* uv)
{
    const Vec qo = mk(q[0], q[1], q[2]), qn = mk(q[9], q[10], q[11]);
    const Vec invU = mk(q[12], q[13], q[14]), invV = mk(q[15], q[16], q[17]);
    const double denom = dot(qn, ray.d);
    if (dabs(denom) <= kEps)
    {
        return false;
    }
    const Vec toOrigin = qo - ray.o;
    const double t = dot(toOrigin, qn) / denom;
    if (t <= 1e-9)
    {
        return false;
    }
    const Vec pos = ray.o + (t * ray.d);
    const Vec local = pos - qo;
    const double u = dot(local, invU), v = dot(local, invV);
    if (u < 0.0 || u > 1.0 || v < 0.0 || v > 1.0)
    {
        return false;
    }
    *position = pos;
    *normal = denom > 0.0 ? neg(qn) : qn;
    *uv = mk(u, v, 0.0);
    *distance = magnitude(pos - ray.o);
    return true;
}

// Quad::update, src/Quad.cpp:21-48: fills q[9..17] from q[0..8].
RT_DEV void quadUpdate(double* q)
{
    const Vec eu = mk(q[3], q[4], q[5]), ev = mk(q[6], q[7], q[8]);
    const Vec n = normalized(cross(eu, ev));
    const double g11 = dot(eu, eu), g22 = dot(ev, ev), g12 = dot(eu, ev);
    const double det = g11 * g22 - g12 * g12;
    Vec iu = mk(0, 0, 0), iv = mk(0, 0, 0);
    if (det != 0.0)
    {
        const double invDet = 1.0 / det;
        iu = (g22 * eu - g12 * ev) * invDet;
        iv = (g11 * ev - g12 * eu) * invDet;
    }
    q[9] = n.x; q[10] = n.y; q[11] = n.z;
    q[12] = iu.x; q[13] = iu.y; q[14] = iu.z;
    q[15] = iv.x; q[16] = iv.y; q[17] = iv.z;
}

// The scene's arrays are never written while a kernel runs.  Reading an element that every lane
// of the wave wants (volume v of the loop over volumes, the first light, ...) through the
// constant address space lets the compiler fetch it once per wave through the scalar cache
// instead of once per lane through the vector path.
template <class T>
RT_DEV const __attribute__((address_space(4))) T* uniformRead(const T* p)
{
    return reinterpret_cast<const __attribute__((address_space(4))) T*>(reinterpret_cast<unsigned long long>(p));
}

// ------------------------------------------------------------------ scene cast
// Per-lane traversal stack in LDS: entry k of lane t lives at base[k * stride] so that
// the 64 lanes of a wave always touch 64 different banks.  An entry packs the node
// index (high word) and the ray parameter at which the ray enters the node's box,
// rounded DOWN to a float (low word) — rounding down can only make pruning rarer.
struct TraversalStack
{
    unsigned long long* base;
    int32_t stride;
};

RT_DEV TraversalStack laneStack()
{
    extern __shared__ unsigned long long rtTraversalStack[];
    return TraversalStack{rtTraversalStack + threadIdx.x, static_cast<int32_t>(blockDim.x)};
}

// The scene cache of the trace kernel lies behind the block's stacks.
RT_DEV unsigned char* sceneCache(const SceneView& s)
{
    extern __shared__ unsigned long long rtTraversalStack[];
    return reinterpret_cast<unsigned char*>(rtTraversalStack + static_cast<uint32_t>(s.stackDepth) * blockDim.x);
}

// Copies SceneView::cacheImage into LDS (every thread of the block takes part; ends with a barrier).
RT_DEV void fillSceneCache(const SceneView& s)
{
    uint4* dst = reinterpret_cast<uint4*>(sceneCache(s));
    const uint4* src = reinterpret_cast<const uint4*>(s.cacheImage);
    for (uint32_t i = threadIdx.x; i < s.cacheBytes / 16; i += blockDim.x)
    {
        dst[i] = src[i];
    }
    __syncthreads();
}

// Slab test of one child box of a node, in single precision with packed operations.  The
// boxes are grown (on the host) by far more than single precision loses on coordinates of the
// scene's size, so that every parameter interval computed here CONTAINS the true one: a box
// the ray meets is never reported as missed, and the entry parameter is never over-estimated.
// Which hit is reported does not depend on these boxes (see castMesh).
typedef float Float2 __attribute__((ext_vector_type(2)));

struct RayPrepSingle
{
    // origin and reciprocal direction paired the way a box's six planes lie in memory:
    // (lo x, lo y) (lo z, hi x) (hi y, hi z)
    Float2 oA, oB, oC, dA, dB, dC;
};

RT_DEV RayPrepSingle prepareRaySingle(const Ray& ray)
{
    float o[3], d[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
    {
        o[i] = static_cast<float>(axisOf(ray.o, i));
        // The hardware's reciprocal (one unit in the last place) of the rounded direction: a few
        // parts in ten million of the parameter, which the boxes' padding covers with room to spare.
        // An axis the ray runs parallel to: a huge finite reciprocal keeps 0 * reciprocal at 0, so
        // an origin between (or on) the planes gives (-huge, +huge) and one outside an empty interval
        const float r = __builtin_amdgcn_rcpf(static_cast<float>(axisOf(ray.d, i)));
        d[i] = __builtin_fminf(__builtin_fmaxf(r, -1e30f), 1e30f);
    }
    RayPrepSingle q;
    q.oA = Float2{o[0], o[1]};
    q.oB = Float2{o[2], o[0]};
    q.oC = Float2{o[1], o[2]};
    q.dA = Float2{d[0], d[1]};
    q.dB = Float2{d[2], d[0]};
    q.dC = Float2{d[1], d[2]};
    return q;
}

// 1 / |direction| in single precision, never too small (2e-6 above what the conversions, the sum and
// the hardware's reciprocal square root can lose): turns a distance bound into a parameter bound
// that never prunes too much.
RT_DEV float reciprocalLengthUp(const Vec& d)
{
    const float x = static_cast<float>(d.x), y = static_cast<float>(d.y), z = static_cast<float>(d.z);
    return __builtin_amdgcn_rsqf(x * x + y * y + z * z) * (1.0f + 2e-6f);
}

RT_DEV bool slabTestChild(const RayPrepSingle& q, const float* box, float* entry)
{
    const Float2 tA = (Float2{box[0], box[1]} - q.oA) * q.dA;  // lo x, lo y
    const Float2 tB = (Float2{box[2], box[3]} - q.oB) * q.dB;  // lo z, hi x
    const Float2 tC = (Float2{box[4], box[5]} - q.oC) * q.dC;  // hi y, hi z
    const float nearX = __builtin_fminf(tA.x, tB.y), farX = __builtin_fmaxf(tA.x, tB.y);
    const float nearY = __builtin_fminf(tA.y, tC.x), farY = __builtin_fmaxf(tA.y, tC.x);
    const float nearZ = __builtin_fminf(tB.x, tC.y), farZ = __builtin_fmaxf(tB.x, tC.y);
    const float tmin = __builtin_fmaxf(0.0f, __builtin_fmaxf(nearX, __builtin_fmaxf(nearY, nearZ)));
    const float tmax = __builtin_fminf(farX, __builtin_fminf(farY, farZ));
    *entry = tmin;
    return !(tmin > tmax);
}

// Closest hit of one mesh, equal to Tree::castRay's exhaustive minimum
// (src/Tree.cpp:133-152,207-237).  The reference visits every node whose box (and
// whose ancestors' boxes) the ray meets, in page / left / right order, and keeps a
// candidate only when it is strictly closer, so of equally distant hits the one
// visited first wins.  Every triangle carries its rank in that visiting order and the
// box of the reference node that holds it (SceneView::triRec), so the same winner is
// "of the triangles the reference would test, the one at the smallest distance, then of
// the smallest rank" — which depends neither on the order of the walk nor on the shape of
// the tree, as long as the walk reaches every such triangle.  That frees the walk to use
// its own tree, to go near child first and to leave out any subtree the ray enters beyond
// the best distance so far (or beyond `bound`, the distance another volume has already
// achieved): such a subtree cannot hold a hit that wins.  The comparison carries a margin
// far above the rounding of the box parameter and of the reconstructed hit distance, so
// ties and near-ties are always visited.  A step reads one 64-byte node — both children's
// boxes and references — and a leaf is a reference, not a node, so its triangles follow
// without another fetch.  Triangle tests are deferred until every lane of the wave has
// either reached a leaf or run out of nodes, so that the lanes test triangles together
// instead of one lane at a time.
// Cached: the kernel holds the top of the tree (ldsNodeCount nodes at ldsNodes) and possibly the
// mesh's triangles (ldsTris, 10 doubles each; null: none) in LDS.
template <bool Cached>
RT_DEV bool castMesh(const SceneView& s, int32_t nodeBase, int32_t nodeCount, int64_t triBase, double extent, double floor,
                     const Ray& ray, const RayPrep& prep, const TraversalStack& stack, double bound, double* bary,
                     Vec* position, double* distance, int32_t* triSlot, int32_t ldsNodeCount = 0,
                     const unsigned char* ldsNodes = nullptr, const double* ldsTris = nullptr)
{
    if (nodeCount <= 0)
    {
        return false;
    }
    RT_DIAG_TRIP(0);  // mesh casts
    const Node* nodes = s.nodes + nodeBase;
    const double* triRec = s.triRec + 16 * triBase;
    const float perLength = reciprocalLengthUp(ray.d);
    const double slack = 1e-11 * (extent + dabs(ray.o.x) + dabs(ray.o.y) + dabs(ray.o.z));
    bool found = false;
    double best = kMaxDouble;
    int32_t bestSlot = 0x7fffffff, bestRank = 0x7fffffff;
    double cutoff = bound * (1.0 + 1e-9) + slack;  // kMaxDouble -> +inf: nothing is pruned
    // the same bound as a ray parameter in single precision, rounded up (never prunes too much)
    float cutoffT = __double2float_ru(cutoff) * perLength;
    const RayPrepSingle single = prepareRaySingle(ray);

    constexpr int32_t kDone = kNoChild;    // nothing left to visit
    constexpr int32_t kPop = kNoChild + 1;  // take the next entry of the stack
    int32_t cur = 0;                        // >= 0 inner node, kDone, kPop, else a leaf reference
    int32_t sp = 0;
    while (true)
    {
        RT_DIAG_TRIP(2);  // outer rounds
        RT_DIAG_TIC(nodeLoopStart);
        while (true)
        {
            RT_DIAG_TRIP(4);  // node steps
            if (cur >= 0)
            {
                // Photons that left the light together mostly stand on the same node: then the
                // node comes through the scalar cache, once for the wave.
                float leftBox[6], rightBox[6];
                int32_t left, right;
                const int32_t shared = __builtin_amdgcn_readfirstlane(cur);
                if (__ballot(cur != shared) == 0)
                {
                    const auto node = uniformRead(nodes + shared);
#pragma unroll
                    for (int k = 0; k < 6; ++k)
                    {
                        leftBox[k] = node->leftBox[k];
                        rightBox[k] = node->rightBox[k];
                    }
                    left = node->left;
                    right = node->right;
                }
                else
                {
                    float4 q0, q1, q2;
                    int4 links;
                    if (Cached && cur < ldsNodeCount)
                    {
                        const float4* node = reinterpret_cast<const float4*>(ldsNodes + 64u * static_cast<uint32_t>(cur));
                        q0 = node[0];
                        q1 = node[1];
                        q2 = node[2];
                        links = reinterpret_cast<const int4*>(node)[3];
                    }
                    else
                    {
                        const Node& node = nodes[cur];
                        q0 = reinterpret_cast<const float4*>(&node)[0];
                        q1 = reinterpret_cast<const float4*>(&node)[1];
                        q2 = reinterpret_cast<const float4*>(&node)[2];
                        links = reinterpret_cast<const int4*>(&node)[3];
                    }
                    leftBox[0] = q0.x; leftBox[1] = q0.y; leftBox[2] = q0.z; leftBox[3] = q0.w; leftBox[4] = q1.x; leftBox[5] = q1.y;
                    rightBox[0] = q1.z; rightBox[1] = q1.w; rightBox[2] = q2.x; rightBox[3] = q2.y; rightBox[4] = q2.z; rightBox[5] = q2.w;
                    left = links.x;
                    right = links.y;
                }
                float el, er;
                bool okl = slabTestChild(single, leftBox, &el);
                bool okr = slabTestChild(single, rightBox, &er);
                okl = okl && left != kNoChild && !(el > cutoffT);
                okr = okr && right != kNoChild && !(er > cutoffT);
                if (okl && okr)
                {
                    const bool leftNear = el <= er;
                    const int32_t farChild = leftNear ? right : left;
                    const float farEntry = leftNear ? er : el;
                    stack.base[sp * stack.stride] =
                        (static_cast<unsigned long long>(static_cast<uint32_t>(farChild)) << 32) | __float_as_uint(farEntry);
                    sp += 1;
                    cur = leftNear ? left : right;
                }
                else
                {
                    cur = okl ? left : (okr ? right : kPop);
                }
            }
            if (cur == kPop)
            {
                cur = kDone;
                while (sp > 0)
                {
                    sp -= 1;
                    const unsigned long long e = stack.base[sp * stack.stride];
                    if (!(__uint_as_float(static_cast<uint32_t>(e)) > cutoffT))
                    {
                        cur = static_cast<int32_t>(e >> 32);
                        break;
                    }
                }
            }
            if (cur < 0)
            {
                break;  // a leaf to test, or done
            }
        }
        RT_DIAG_TOC(nodeLoopStart, 2);
        if (cur == kDone)
        {
            break;
        }
        const int32_t packed = -1 - cur;
        const int32_t pendingStart = packed >> 4;
        const int32_t pendingEnd = pendingStart + (packed & 15) + 1;
        RT_DIAG_TIC(triLoopStart);
        for (int32_t t = pendingStart; t < pendingEnd; ++t)
        {
            RT_DIAG_TRIP(6);  // triangle tests
            double b3[3], d;
            Vec pos;
            const double* rec = triRec + 16 * t;
            // rank and admission weight travel with the vertices: same 128-byte line, same wait
            unsigned long long word;
            bool hitsTriangle;
            if (Cached && ldsTris != nullptr)
            {
                const double* cached = ldsTris + 10 * t;
                word = reinterpret_cast<const unsigned long long*>(cached)[9];
                hitsTriangle = rayIntersectsTriangle(ray, prep, cached, b3, &pos, &d);
            }
            else
            {
                word = reinterpret_cast<const unsigned long long*>(rec)[9];
                hitsTriangle = rayIntersectsTriangle(ray, prep, rec, b3, &pos, &d);
            }
            // (floor: the tree of several flat volumes takes no hit at or below the self-hit floor, which
            // drops such a volume's only hit as the caller would; every other tree passes -1, no distance is)
            if (hitsTriangle && d > floor)
            {
                const int32_t rank = static_cast<int32_t>(static_cast<uint32_t>(word));
                bool wins = !found || d < best || (d == best && rank < bestRank);
                if (wins)
                {
                    // The reference tests a triangle only when the ray meets its own node's box
                    // (and with it every ancestor's).  Away from the triangle's edges that is
                    // certain; otherwise, or when the ray runs parallel to an axis, ask the box (in
                    // double precision, as the reference does: the three reciprocals that takes are
                    // worked out here, for the few rays that come this way).
                    const double certain = static_cast<double>(__uint_as_float(static_cast<uint32_t>(word >> 32)));
                    const double least = dmin(b3[0], dmin(b3[1], b3[2]));
                    if (prep.parallel || !(least >= certain))
                    {
                        double ignored;
                        wins = slabTest(ray, prepareSlab(ray), rec + 10, &ignored);
                    }
                }
                if (wins)
                {
                    found = true;
                    best = d;
                    bestSlot = t;
                    bestRank = rank;
                    bary[0] = b3[0];
                    bary[1] = b3[1];
                    bary[2] = b3[2];
                    *position = pos;
                    cutoff = dmin(best, bound) * (1.0 + 1e-9) + slack;
                    cutoffT = __double2float_ru(cutoff) * perLength;
                }
            }
        }
        RT_DIAG_TOC(triLoopStart, 3);
        cur = kPop;
    }
    *triSlot = bestSlot;
    *distance = best;
    return found;
}

// ------------------------------------------------------------------ keyframed poses
// Property<T>::evaluate, reference include/Property.h:95-213: cubic Hermite through the
// keyframes with Catmull-Rom tangents where none are given (:161-186), held outside the
// keyed range.  `dims` is 1 for the angle track and 3 for the position track.
RT_DEV double autoTangent(const rt_keyframe* keys, int64_t n, int64_t i, int d)
{
    if (n < 2) return 0.0;
    int64_t a = i - 1, b = i + 1;
    if (i == 0)
    {
        a = 0;
        b = 1;
    }
    else if (i == n - 1)
    {
        a = n - 2;
        b = n - 1;
    }
    const double span = keys[b].time - keys[a].time;
    if (span <= 0.0) return 0.0;
    return (keys[b].value[d] - keys[a].value[d]) * (1.0 / span);
}

RT_DEV void evaluateTrack(const rt_keyframe* keys, int64_t n, int dims, double t, double* out)
{
    int64_t held = -1;
    int64_t idx = 1;
    if (n == 1 || t <= keys[0].time)
    {
        held = 0;
    }
    else if (t >= keys[n - 1].time)
    {
        held = n - 1;
    }
    else
    {
        while (idx < n && keys[idx].time <= t)
        {
            ++idx;
        }
        if (keys[idx].time - keys[idx - 1].time <= 0.0)
        {
            held = idx - 1;
        }
    }
    if (held >= 0)
    {
        for (int d = 0; d < dims; ++d) out[d] = keys[held].value[d];
        return;
    }
    const rt_keyframe& k0 = keys[idx - 1];
    const rt_keyframe& k1 = keys[idx];
    const double dt = k1.time - k0.time;
    const double u = (t - k0.time) / dt;
    const double u2 = u * u;
    const double u3 = u2 * u;
    const double h00 = 2.0 * u3 - 3.0 * u2 + 1.0;
    const double h10 = u3 - 2.0 * u2 + u;
    const double h01 = -2.0 * u3 + 3.0 * u2;
    const double h11 = u3 - u2;
    for (int d = 0; d < dims; ++d)
    {
        const double m0 = k0.use_auto_tangent ? autoTangent(keys, n, idx - 1, d) : k0.out_tangent[d];
        const double m1 = k1.use_auto_tangent ? autoTangent(keys, n, idx, d) : k1.in_tangent[d];
        out[d] = k0.value[d] * h00 + m0 * (h10 * dt) + k1.value[d] * h01 + m1 * (h11 * dt);
    }
}

struct Pose
{
    Vec position;
    Quat rotation;
};

// KeyframedAnimationQuery::transformAt, reference include/AnimationQuery.h:130-158, with
// Quaternion::fromAxisAngle (src/Quaternion.cpp:76-86) on the shared sin / cos.
RT_DEV Pose poseAt(const SceneView& s, int32_t animation, float time)
{
    const rt_animation& a = s.animations[animation];
    const double t = static_cast<double>(time);
    Pose pose;
    pose.position = mk(a.constant_position[0], a.constant_position[1], a.constant_position[2]);
    if (a.position_key_count > 0)
    {
        double p[3];
        evaluateTrack(s.keyframes + a.first_position_key, a.position_key_count, 3, t, p);
        pose.position = mk(p[0], p[1], p[2]);
    }
    const Quat base{a.base_rotation[0], a.base_rotation[1], a.base_rotation[2], a.base_rotation[3]};
    pose.rotation = base;
    if (a.has_rotation)
    {
        double angle[1] = {0.0};  // an empty track is the constant 0
        if (a.rotation_key_count > 0)
        {
            evaluateTrack(s.keyframes + a.first_rotation_key, a.rotation_key_count, 1, t, angle);
        }
        const double half = angle[0] / 2.0;
        const double sa = rt_sin(half);
        const double ca = rt_cos(half);
        const Quat spin{a.rotation_axis[0] * sa, a.rotation_axis[1] * sa, a.rotation_axis[2] * sa, ca};
        pose.rotation = qmul(spin, base);
    }
    return pose;
}

RT_DEV Quat quatInverse(const Quat& q)  // reference src/Quaternion.cpp:39-52
{
    const double m = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const double scale = 1.0 / m;
    return Quat{-q.x * scale, -q.y * scale, -q.z * scale, q.w * scale};
}

// The reference keeps a volume's hit when it is strictly nearer than the nearest so far, and visits the
// volumes in their order: of equally distant hits the first volume's stays.  Flat meshes of one pose are
// cast together at the place of the first of them, so the order of the visits is no longer the order of the
// volumes and the tie is settled by the volumes' numbers instead.
// (`groups`: the scene has such casts; without them the visits are in the volumes' order, as the reference's)
RT_DEV bool closer(bool groups, double dist, int32_t volume, double nearest, int32_t nearestVolume)
{
    return dist < nearest || (groups && dist == nearest && volume < nearestVolume);
}

// Nearest valid hit over all volumes: Volume::castRayAt for each volume in object
// order (src/Volume.cpp:34-81), then the strict-minimum selection with the self-hit
// floor applied AFTER each volume's own closest hit was chosen
// (src/Worker.cpp:556-567).  When `needSurface` is false only the distance is
// produced (occlusion test, src/Worker.cpp:314-333).  With Animated set, a keyframed
// volume is posed at `time` (the photon's or the camera sample's) for the cast; without
// it every volume keeps its scene-load transform and `time` is unused.
// Grouped: the scene may hold casts of several flat meshes at once (kVolumeMeshGroup); a kernel that is
// only launched for scenes without them passes false and carries none of their handling.
template <bool Animated, bool Cached = false, bool Grouped = true>
RT_DEV bool castScene(const SceneView& s, const Ray& ray, float time, double selfHit, bool needSurface, Hit* out)
{
    const TraversalStack stack = laneStack();
    double minDistance = kMaxDouble;
    bool valid = false;
    Vec bestPos = mk(0, 0, 0), bestNrm = mk(0, 0, 0);
    int32_t bestVolume = -1, bestTri = -1;
    // the per-volume box test runs in single precision against padded boxes, as in castMesh
    const float perWorldLength = reciprocalLengthUp(ray.d);
    Ray local;
    local.o = local.d = mk(0, 0, 0);
    RayPrep prep = {};
    int32_t heldPose = -1;  // Volume::poseId of the pose `local` is in (none yet)
    bool prepared = false;  // `prep` belongs to `local`
    for (int32_t v = 0; v < s.volumeCount; ++v)
    {
        RT_DIAG_TIC(boxStart);
        const auto vol = uniformRead(s.volumes + v);  // v is the same in every lane
        const int32_t type = vol->type;
        if (Grouped && type == kVolumePassedOver)
        {
            continue;  // cast with the first flat mesh of its pose
        }
        const bool group = Grouped && type == kVolumeMeshGroup;
        if (vol->hasWorldBox)
        {
            float box[6], entry;
#pragma unroll
            for (int k = 0; k < 6; ++k)
            {
                box[k] = vol->worldBox[k];
            }
            // The world ray's single-precision form is worked out again for every volume (a dozen
            // instructions): kept across the casts it would hold twelve registers the kernel does
            // not have.  The empty statement keeps the compiler from keeping it all the same.
            Ray here = ray;
            asm volatile("" : "+v"(here.d.x), "+v"(here.d.y), "+v"(here.d.z));
            const RayPrepSingle worldSingle = prepareRaySingle(here);
            // missed, or entered beyond the closest hit so far (as a parameter bound rounded up)
            if (!slabTestChild(worldSingle, box, &entry) ||
                entry > __double2float_ru(minDistance * (1.0 + 1e-9)) * perWorldLength)
            {
                RT_DIAG_TOC(boxStart, 7);
                continue;
            }
        }
        RT_DIAG_TOC(boxStart, 7);
        RT_DIAG_TIC(poseStart);
        // the ray in the volume's object space, and what the mesh tests derive from it: kept while
        // the volumes that follow share the pose (the walls of a room are one pose)
        if (vol->poseId != heldPose)
        {
            Vec vpos = mk(vol->pos[0], vol->pos[1], vol->pos[2]);
            Quat inv = Quat{vol->inv.x, vol->inv.y, vol->inv.z, vol->inv.w};
            if (Animated && vol->animation >= 0)
            {
                const Pose pose = poseAt(s, vol->animation, time);
                vpos = pose.position;
                inv = quatInverse(pose.rotation);
            }
            local.o = rotate(inv, ray.o - vpos);
            local.d = rotate(inv, ray.d);
            heldPose = vol->poseId;
            prepared = false;
        }
        Vec pos, nrm = mk(0, 0, 0);
        double dist;
        bool found = false;
        int32_t tri = -1, hitVolume = v;
        if (type == RT_VOLUME_MESH || group)
        {
            if (!prepared)
            {
                prep = prepareRay(local);
                prepared = true;
            }
            RT_DIAG_TOC(poseStart, 8);
            double bary[3];
            int32_t slot = 0;
            const int64_t triBase = vol->triBase;
            if (Cached)
            {
                const unsigned char* cache = sceneCache(s);
                const int32_t triOffset = vol->ldsTriOffset;
                found = castMesh<true>(s, vol->nodeBase, vol->nodeCount, triBase, vol->extent, group ? selfHit : -1.0, local, prep, stack, minDistance,
                                       bary, &pos, &dist, &slot, vol->ldsNodeCount, cache + vol->ldsNodeOffset,
                                       triOffset >= 0 ? reinterpret_cast<const double*>(cache + triOffset) : nullptr);
            }
            else
            {
                found = castMesh<false>(s, vol->nodeBase, vol->nodeCount, triBase, vol->extent, group ? selfHit : -1.0, local, prep, stack, minDistance,
                                        bary, &pos, &dist, &slot);
            }
            RT_DIAG_TIC(normalStart);
            if (group && found)
            {
                hitVolume = s.triOwner[triBase + slot];
            }
            if (found && closer(Grouped && s.groupCount > 0, dist, hitVolume, minDistance, bestVolume) && dist > selfHit)
            {
                if (needSurface)
                {
                    nrm = triangleNormal(s.triNrm + 9 * (triBase + slot), bary);
                    tri = s.triId[triBase + slot];
                }
            }
            else
            {
                found = false;
            }
            RT_DIAG_TOC(normalStart, 9);
        }
        else if (type == RT_VOLUME_SPHERE)
        {
            found = rayIntersectsSphere(local, mk(vol->param[0], vol->param[1], vol->param[2]), vol->param[3], &pos, &nrm,
                                        &dist);
        }
        else if (type == RT_VOLUME_QUAD)
        {
            double quad[18];
#pragma unroll
            for (int k = 0; k < 18; ++k)
            {
                quad[k] = vol->param[k];
            }
            Vec uv;
            found = rayIntersectsQuad(local, quad, &pos, &nrm, &dist, &uv);
        }
        else
        {
            found = rayIntersectsPlane(local, mk(vol->param[0], vol->param[1], vol->param[2]), vol->param[3], &pos, &nrm,
                                       &dist);
        }
        if (found && closer(Grouped && s.groupCount > 0, dist, hitVolume, minDistance, bestVolume) && dist > selfHit)
        {
            valid = true;
            minDistance = dist;
            bestPos = pos;
            bestNrm = nrm;
            bestVolume = hitVolume;
            bestTri = tri;
        }
    }
    out->distance = minDistance;
    RT_DIAG_TIC(worldStart);
    if (valid && needSurface)
    {
        const Volume& vol = s.volumes[bestVolume];
        Vec vpos = mk(vol.pos[0], vol.pos[1], vol.pos[2]);
        Quat rot = vol.rot;
        if (Animated && vol.animation >= 0)
        {
            const Pose pose = poseAt(s, vol.animation, time);  // the pose the winning cast used
            vpos = pose.position;
            rot = pose.rotation;
        }
        out->position = vpos + rotate(rot, bestPos);
        out->normal = rotate(rot, bestNrm);
        out->material = vol.material;
        out->volume = bestVolume;
        out->triangle = bestTri;
    }
    RT_DIAG_TOC(worldStart, 10);
    return valid;
}

// ------------------------------------------------------------------ random draws
// Draws come either from the per-photon Philox stream or from an explicit tape
// (function-level entry points).
struct Rng
{
    rt_philox_state philox;
    const double* tape;
    int next;
    uint32_t seed;
    RT_DEV void begin(uint32_t light, uint64_t photon, uint32_t stage)
    {
        if (!tape)
        {
            rt_philox_begin(&philox, seed, light, photon, stage);
        }
    }
    RT_DEV double value(double scale = 1.0)
    {
        if (tape)
        {
            return tape[next++] * scale;
        }
        return rt_philox_next(&philox) * scale;
    }
};

// ------------------------------------------------------------------ BSDFs
struct BSDFSample
{
    Vec direction;
    Color weight;
    double pdf;
    bool isDelta, valid;
};

RT_DEV bool materialIsDelta(int32_t type) { return type == RT_MATERIAL_MIRROR || type == RT_MATERIAL_DIELECTRIC; }

// src/LambertianMaterial.cpp:13-25
RT_DEV void buildBasis(const Vec& n, Vec* tangent, Vec* bitangent)
{
    if (n.z < -0.9999999)
    {
        *tangent = mk(0.0, -1.0, 0.0);
        *bitangent = mk(-1.0, 0.0, 0.0);
        return;
    }
    const double a = 1.0 / (1.0 + n.z);
    const double b = -n.x * n.y * a;
    *tangent = mk(1.0 - n.x * n.x * a, b, -n.x);
    *bitangent = mk(b, 1.0 - n.y * n.y * a, -n.y);
}

RT_DEV double clamp01(double v) { return v < 0.0 ? 0.0 : (1.0 < v ? 1.0 : v); }  // std::clamp(v, 0, 1)

// src/MicrofacetMaterial.cpp:29-77
RT_DEV double ggxD(double cosThetaH, double alpha)
{
    if (cosThetaH <= 0.0) return 0.0;
    const double a2 = alpha * alpha;
    const double cos2 = cosThetaH * cosThetaH;
    const double denom = (a2 - 1.0) * cos2 + 1.0;
    return a2 / (kPi * denom * denom);
}
RT_DEV double smithLambda(double cosTheta, double alpha)
{
    if (cosTheta >= 1.0 || cosTheta <= 0.0) return 0.0;
    const double cos2 = cosTheta * cosTheta;
    const double sin2 = dmax(0.0, 1.0 - cos2);
    const double tan2 = sin2 / cos2;
    return 0.5 * (__builtin_sqrt(1.0 + alpha * alpha * tan2) - 1.0);
}
RT_DEV double smithG1(double cosTheta, double alpha) { return 1.0 / (1.0 + smithLambda(cosTheta, alpha)); }
RT_DEV Color schlickFresnel(const float* F0, double cosTheta)
{
    const double t = dmax(0.0, 1.0 - cosTheta);
    const double t5 = t * t * t * t * t;
    return Color{static_cast<float>(F0[0] + (1.0 - F0[0]) * t5), static_cast<float>(F0[1] + (1.0 - F0[1]) * t5),
                 static_cast<float>(F0[2] + (1.0 - F0[2]) * t5)};
}

// Material::sample for the four materials: src/LambertianMaterial.cpp:52-75,
// src/MirrorMaterial.cpp:15-27, src/DielectricMaterial.cpp:30-136,
// src/MicrofacetMaterial.cpp:93-123,141-214.
RT_DEV BSDFSample sampleMaterial(const Material& m, const Vec& incident, const Vec& normal, Rng& rng)
{
    BSDFSample s;
    s.direction = mk(0, 0, 0);
    s.weight = Color{1.0f, 1.0f, 1.0f};
    s.pdf = 0.0;
    s.isDelta = false;
    s.valid = false;
    const Color albedo{m.color[0], m.color[1], m.color[2]};
    if (m.type == RT_MATERIAL_LAMBERTIAN)
    {
        Vec tangent, bitangent;
        buildBasis(normal, &tangent, &bitangent);
        const double u1 = rng.value();
        const double u2 = rng.value();
        const double r = __builtin_sqrt(u1);
        const double phi = kPi2 * u2;
        double sn, cs;
        rt_sincos(phi, &sn, &cs);
        const Vec local = mk(r * cs, r * sn, __builtin_sqrt(dmax(0.0, 1.0 - u1)));
        Vec world = tangent * local.x + bitangent * local.y + normal * local.z;
        world = normalized(world);
        const double cosTheta = dmax(0.0, dot(world, normal));
        s.direction = world;
        s.weight = albedo;
        s.pdf = cosTheta / kPi;
        s.valid = cosTheta > 0.0;
        return s;
    }
    if (m.type == RT_MATERIAL_MIRROR)
    {
        s.direction = normalized(reflected(incident, normal));
        s.weight = albedo;
        s.pdf = 1.0;
        s.isDelta = true;
        s.valid = dot(s.direction, normal) > 0.0;
        return s;
    }
    if (m.type == RT_MATERIAL_DIELECTRIC)
    {
        const Vec dir = normalized(incident);
        Vec n = normalized(normal);
        double cosI = dot(dir, n);
        double n1, n2;
        if (cosI < 0.0)
        {
            n1 = 1.0;
            n2 = m.ior;
            cosI = -cosI;
        }
        else
        {
            n = neg(n);
            n1 = m.ior;
            n2 = 1.0;
        }
        const Vec reflectDir = normalized(reflected(dir, n));
        Vec refractDir = reflectDir;
        bool tir = false;
        double reflectance = 1.0;
        const double eta = n1 / n2;
        const double sinT2 = eta * eta * (1.0 - cosI * cosI);
        if (sinT2 > 1.0)
        {
            tir = true;
        }
        else
        {
            const double cosT = __builtin_sqrt(dmax(0.0, 1.0 - sinT2));
            refractDir = normalized(dir * eta + n * (eta * cosI - cosT));
            const double cosForFresnel = (n1 <= n2) ? cosI : cosT;
            double r0 = (n1 - n2) / (n1 + n2);
            r0 = r0 * r0;
            const double mm = clamp01(1.0 - cosForFresnel);
            const double m5 = mm * mm * mm * mm * mm;
            reflectance = clamp01(r0 + (1.0 - r0) * m5);
        }
        s.isDelta = true;
        s.pdf = 1.0;
        const double xi = rng.value();
        s.direction = (tir || xi < reflectance) ? reflectDir : refractDir;
        s.weight = albedo;
        s.valid = dot(s.direction, s.direction) > 0.5;
        return s;
    }
    // Microfacet (GGX, VNDF sampling)
    const Vec wi = neg(incident);
    const double cosThetaI = dot(wi, normal);
    if (cosThetaI <= 0.0)
    {
        return s;
    }
    Vec tangent, bitangent;
    buildBasis(normal, &tangent, &bitangent);
    const Vec Ve = mk(dot(wi, tangent), dot(wi, bitangent), cosThetaI);
    const double u1 = rng.value();
    const double u2 = rng.value();
    const double alpha = m.alpha;
    Vec hLocal;
    {
        const Vec Vh = normalized(mk(alpha * Ve.x, alpha * Ve.y, Ve.z));
        const double lensq = Vh.x * Vh.x + Vh.y * Vh.y;
        const Vec T1 = (lensq > 0.0) ? mk(-Vh.y, Vh.x, 0.0) * (1.0 / __builtin_sqrt(lensq)) : mk(1.0, 0.0, 0.0);
        const Vec T2 = cross(Vh, T1);
        const double r = __builtin_sqrt(u1);
        const double phi = kPi2 * u2;
        double sn, cs;
        rt_sincos(phi, &sn, &cs);
        const double t1 = r * cs;
        double t2 = r * sn;
        const double sh = 0.5 * (1.0 + Vh.z);
        t2 = (1.0 - sh) * __builtin_sqrt(dmax(0.0, 1.0 - t1 * t1)) + sh * t2;
        const double t3 = __builtin_sqrt(dmax(0.0, 1.0 - t1 * t1 - t2 * t2));
        const Vec Nh = T1 * t1 + T2 * t2 + Vh * t3;
        hLocal = normalized(mk(alpha * Nh.x, alpha * Nh.y, dmax(0.0, Nh.z)));
    }
    Vec wh = tangent * hLocal.x + bitangent * hLocal.y + normal * hLocal.z;
    wh = normalized(wh);
    const double wiDotWh = dot(wi, wh);
    if (wiDotWh <= 0.0)
    {
        return s;
    }
    Vec wo = wh * (2.0 * wiDotWh) - wi;
    wo = normalized(wo);
    const double cosThetaO = dot(wo, normal);
    const double cosThetaH = dot(wh, normal);
    if (cosThetaO <= 0.0 || cosThetaH <= 0.0)
    {
        return s;
    }
    const double D = ggxD(cosThetaH, alpha);
    const double G1i = smithG1(cosThetaI, alpha);
    const double G2 = smithG1(cosThetaI, alpha) * smithG1(cosThetaO, alpha);
    const Color F = schlickFresnel(m.color, wiDotWh);
    const double weightScalar = (G1i > 0.0) ? (G2 / G1i) : 0.0;
    s.direction = wo;
    s.weight = F * static_cast<float>(weightScalar);
    s.pdf = (G1i * D) / (4.0 * cosThetaI);
    s.valid = true;
    return s;
}

// Material::evaluate: src/LambertianMaterial.cpp:94-113, src/MicrofacetMaterial.cpp:258-286;
// delta materials are black.
RT_DEV Color evaluateMaterial(const Material& m, const Vec& wi, const Vec& wo, const Vec& normal)
{
    const Color black{0.0f, 0.0f, 0.0f};
    if (m.type == RT_MATERIAL_LAMBERTIAN)
    {
        if (dot(wi, normal) <= 0.0 || dot(wo, normal) <= 0.0)
        {
            return black;
        }
        return Color{m.color[0], m.color[1], m.color[2]} * static_cast<float>(1.0 / kPi);
    }
    if (m.type == RT_MATERIAL_MICROFACET)
    {
        const double ci = dot(wi, normal), co = dot(wo, normal);
        if (ci <= 0.0 || co <= 0.0) return black;
        Vec wh = wi + wo;
        if (dot(wh, wh) < 1e-20) return black;
        wh = normalized(wh);
        const double ch = dot(wh, normal);
        if (ch <= 0.0) return black;
        const double D = ggxD(ch, m.alpha);
        const double G2 = smithG1(ci, m.alpha) * smithG1(co, m.alpha);
        const Color F = schlickFresnel(m.color, dmax(0.0, dot(wi, wh)));
        const double scalar = (D * G2) / (4.0 * ci * co);
        return F * static_cast<float>(scalar);
    }
    return black;
}

// Material::pdf: src/LambertianMaterial.cpp:115-123, src/MicrofacetMaterial.cpp:297-333
RT_DEV double pdfMaterial(const Material& m, const Vec& wi, const Vec& wo, const Vec& normal)
{
    if (m.type == RT_MATERIAL_LAMBERTIAN)
    {
        const double c = dot(wo, normal);
        return c <= 0.0 ? 0.0 : c / kPi;
    }
    if (m.type == RT_MATERIAL_MICROFACET)
    {
        const double ci = dot(wi, normal), co = dot(wo, normal);
        if (ci <= 0.0 || co <= 0.0) return 0.0;
        Vec wh = wi + wo;
        if (dot(wh, wh) < 1e-20) return 0.0;
        wh = normalized(wh);
        const double ch = dot(wh, normal);
        if (ch <= 0.0) return 0.0;
        if (dot(wi, wh) <= 0.0) return 0.0;
        return (smithG1(ci, m.alpha) * ggxD(ch, m.alpha)) / (4.0 * ci);
    }
    return 0.0;
}

// ------------------------------------------------------------------ emission
// src/Vector.cpp:319-329
RT_DEV Vec randomSphere(Rng& rng)
{
    const double theta = 2 * kPi * rng.value();
    const double phi = rt_acos(1.0 - 2.0 * rng.value());
    double sp, cp, st, ct;
    rt_sincos(phi, &sp, &cp);
    rt_sincos(theta, &st, &ct);
    return mk(sp * ct * 1.0, sp * st * 1.0, cp * 1.0);
}

// src/Vector.cpp:314-317; the reference binary draws the magnitude before the sphere
// direction (operand evaluation order of g++ on x86-64, confirmed by the emission
// fixtures).
RT_DEV Vec randomVector(Rng& rng, double magnitude)
{
    const double scale = rng.value(magnitude);
    const Vec sphere = randomSphere(rng);
    return sphere * scale;
}

// Light::emit: src/OmniLight.cpp:23-44, src/SpotLight.cpp:50-70 with
// src/AngleGenerator.cpp:8-36, src/ParallelLight.cpp:26-47, src/AreaLight.cpp:106-155.
RT_DEV Ray emitRay(const Light& l, Rng& rng)
{
    const Vec lpos = mk(l.pos[0], l.pos[1], l.pos[2]);
    const Vec forward = mk(l.forward[0], l.forward[1], l.forward[2]);
    Ray ray;
    if (l.type == RT_LIGHT_OMNI)
    {
        const Vec direction = randomSphere(rng);
        Vec offset = mk(0, 0, 0);
        if (l.innerRadius > 0)
        {
            offset = randomVector(rng, l.innerRadius);
        }
        ray.o = lpos + offset;
        ray.d = direction;
        return ray;
    }
    if (l.type == RT_LIGHT_SPOT)
    {
        const Vec center = forward;
        const double theta = rng.value(kPi2);
        double value = rng.value(1.0);
        value = dmin(dmax(-1.0, value), 1.0);
        const double angle = value * l.maxAngle;  // pow(value, linearity = 1) == value
        const bool useZ = dabs(dot(mk(0, 1, 0), center)) > 0.9;
        const Vec crossAxis = useZ ? mk(0, 0, 1) : mk(0, 1, 0);
        double sa, ca, sh, ch;
        rt_sincos(angle, &sa, &ca);
        rt_sincos(theta / 2.0, &sh, &ch);
        Vec offset = cross(center, crossAxis) * sa;
        const Quat rotation{center.x * sh, center.y * sh, center.z * sh, ch};
        offset = rotate(rotation, offset);
        const Vec offsetDirection = (center * ca) + offset;
        Vec origin = mk(0, 0, 0);
        if (l.innerRadius > 0)
        {
            origin = randomVector(rng, l.innerRadius);
        }
        ray.o = lpos + origin;
        ray.d = offsetDirection;
        return ray;
    }
    if (l.type == RT_LIGHT_PARALLEL)
    {
        Vec offset = mk(0, 0, 0);
        if (l.radius > 0)
        {
            offset = randomVector(rng, l.radius);
            offset = offset - (dot(offset, forward) * forward);
            offset = normalized(offset) * (__builtin_sqrt(magnitude(offset) / l.radius) * l.radius);
        }
        ray.o = lpos + offset;
        ray.d = forward;
        return ray;
    }
    const Vec right = rotate(l.rot, mk(1, 0, 0));
    const Vec up = rotate(l.rot, mk(0, 1, 0));
    const Vec normal = rotate(l.rot, mk(0, 0, 1));
    Vec origin;
    if (l.shape == 1)
    {
        const double theta = rng.value(kPi2);
        const double r = l.radius * __builtin_sqrt(rng.value(1.0));
        double st, ct;
        rt_sincos(theta, &st, &ct);
        origin = lpos + (right * (r * ct)) + (up * (r * st));
    }
    else
    {
        const double u = (rng.value(1.0) - 0.5) * l.width;
        const double v = (rng.value(1.0) - 0.5) * l.height;
        origin = lpos + (right * u) + (up * v);
    }
    const double diskTheta = rng.value(kPi2);
    const double diskR = __builtin_sqrt(rng.value(1.0));
    double sd, cd;
    rt_sincos(diskTheta, &sd, &cd);
    const double dx = diskR * cd;
    const double dy = diskR * sd;
    const double dz = __builtin_sqrt(dmax(0.0, 1.0 - (dx * dx) - (dy * dy)));
    Vec direction = (right * dx) + (up * dy) + (normal * dz);
    direction = normalized(direction);
    ray.o = origin;
    ray.d = direction;
    return ray;
}


// ------------------------------------------------------------------ camera side
// Camera::resolveEyeRotationAt, reference src/Camera.cpp:388-402: the camera with its eye
// and axes at `time` (a static camera is returned as it is).
RT_DEV Camera cameraAt(const SceneView& s, const Camera& cam, float time)
{
    Camera posed = cam;
    if (cam.animation >= 0)
    {
        const Pose pose = poseAt(s, cam.animation, time);
        const Vec right = rotate(pose.rotation, mk(1.0, 0.0, 0.0));
        const Vec up = rotate(pose.rotation, mk(0.0, 1.0, 0.0));
        const Vec forward = rotate(pose.rotation, mk(0.0, 0.0, 1.0));
        posed.pos[0] = pose.position.x; posed.pos[1] = pose.position.y; posed.pos[2] = pose.position.z;
        posed.right[0] = right.x; posed.right[1] = right.y; posed.right[2] = right.z;
        posed.up[0] = up.x; posed.up[1] = up.y; posed.up[2] = up.z;
        posed.forward[0] = forward.x; posed.forward[1] = forward.y; posed.forward[2] = forward.z;
    }
    return posed;
}

// Camera::buildPrimaryRay, reference src/Camera.cpp:404-493, for a camera already posed.
RT_DEV Ray buildPrimaryRay(const Camera& cam, int32_t px, int32_t py, Rng* rng)
{
    double offX = 0.5, offY = 0.5;
    if (rng)
    {
        offX = rng->value(1.0);
        offY = rng->value(1.0);
    }
    const double u = (static_cast<double>(px) + offX) / static_cast<double>(cam.width);
    const double v = (static_cast<double>(py) + offY) / static_cast<double>(cam.height);
    const double sx = 2.0 * u - 1.0, sy = 2.0 * v - 1.0;
    const Vec right = mk(cam.right[0], cam.right[1], cam.right[2]), up = mk(cam.up[0], cam.up[1], cam.up[2]);
    const Vec forward = mk(cam.forward[0], cam.forward[1], cam.forward[2]), eye = mk(cam.pos[0], cam.pos[1], cam.pos[2]);
    Ray ray;
    if (cam.projection == RT_PROJECTION_ORTHOGRAPHIC)
    {
        const double halfH = cam.orthographicHeight / 2.0;
        ray.o = eye + (right * (sx * cam.aspect * halfH)) + (up * (sy * halfH));
        ray.d = normalized(forward);
        return ray;
    }
    if (cam.projection == RT_PROJECTION_REALLENS)
    {
        const Vec pinholeDir =
            normalized(forward + (right * (sx * cam.aspect * cam.halfHeight)) + (up * (sy * cam.halfHeight)));
        const double cosToAxis = dot(pinholeDir, forward);
        const double t = (cosToAxis > 1e-9) ? (cam.focusDistance / cosToAxis) : cam.focusDistance;
        const Vec focusPoint = eye + (pinholeDir * t);
        double lensX = 0.0, lensY = 0.0;
        if (rng)
        {
            const double r = cam.apertureRadius * __builtin_sqrt(rng->value(1.0));
            const double theta = rng->value(kPi2);
            double st, ct;
            rt_sincos(theta, &st, &ct);
            lensX = r * ct;
            lensY = r * st;
        }
        ray.o = eye + (right * lensX) + (up * lensY);
        ray.d = normalized(focusPoint - ray.o);
        return ray;
    }
    ray.o = eye;
    ray.d = normalized(forward + (right * (sx * cam.aspect * cam.halfHeight)) + (up * (sy * cam.halfHeight)));
    return ray;
}

// firstHit + firstEmitterHit + intersectEmitterPatch, reference src/ProbeGather.cpp:57-127,
// src/EmitterPatch.cpp:55-88.  material == -1 marks an emitter face.
RT_DEV bool firstHit(const SceneView& s, const Ray& ray, float time, Hit* closest)
{
    bool found = castScene<true>(s, ray, time, kEps, true, closest);
    double bestT = __builtin_inf();
    int32_t best = -1;
    for (int32_t i = 0; i < s.patchCount; ++i)
    {
        const EmitterPatch& patch = s.patches[i];
        const Vec n = mk(patch.normal[0], patch.normal[1], patch.normal[2]);
        const Vec center = mk(patch.center[0], patch.center[1], patch.center[2]);
        const double denom = dot(ray.d, n);
        if (denom >= -kEps) continue;
        const double t = dot(center - ray.o, n) / denom;
        if (t <= kEps) continue;
        const Vec hitPoint = ray.o + (ray.d * t);
        const Vec local = hitPoint - center;
        const double u = dot(local, mk(patch.right[0], patch.right[1], patch.right[2]));
        const double v = dot(local, mk(patch.up[0], patch.up[1], patch.up[2]));
        if (patch.isDisc)
        {
            if ((u * u + v * v) > (patch.radius * patch.radius)) continue;
        }
        else
        {
            if (dabs(u) > patch.halfWidth || dabs(v) > patch.halfHeight) continue;
        }
        if (t > kEps && t < bestT)
        {
            bestT = t;
            best = i;
        }
    }
    if (best >= 0 && (!found || bestT < closest->distance))
    {
        const EmitterPatch& patch = s.patches[best];
        closest->position = ray.o + ray.d * bestT;
        closest->normal = mk(patch.normal[0], patch.normal[1], patch.normal[2]);
        closest->distance = bestT;
        closest->material = -1;
        closest->volume = -1;
        closest->triangle = -1;
        found = true;
    }
    return found;
}

struct ExtendResult  // reference src/ProbeGather.cpp:138-147
{
    Hit hit;
    Color throughput;
    double unfoldedPathLength;
    Vec finalDirection;
    bool traversedDelta, valid;
};

// extendAndRecord, reference src/ProbeGather.cpp:154-229
RT_DEV ExtendResult extendAndRecord(const SceneView& s, Rng& rng, Ray ray, float time)
{
    ExtendResult out;
    out.throughput = Color{1.0f, 1.0f, 1.0f};
    out.unfoldedPathLength = 0.0;
    out.finalDirection = mk(0, 0, 0);
    out.traversedDelta = false;
    out.valid = false;
    for (int depth = 0; depth < 8; ++depth)
    {
        Hit hit;
        if (!firstHit(s, ray, time, &hit))
        {
            return out;
        }
        out.unfoldedPathLength += hit.distance;
        bool terminal = hit.material < 0;
        if (!terminal)
        {
            terminal = !materialIsDelta(s.materials[hit.material].type);
        }
        if (terminal)
        {
            out.hit = hit;
            out.finalDirection = ray.d;
            out.valid = true;
            return out;
        }
        out.traversedDelta = true;
        const BSDFSample smp = sampleMaterial(s.materials[hit.material], ray.d, hit.normal, rng);
        if (!smp.valid)
        {
            return out;
        }
        out.throughput = out.throughput * smp.weight;
        const Vec nextDir = normalized(smp.direction);
        const double nDotDir = dot(nextDir, hit.normal);
        const Vec offsetNormal = (nDotDir >= 0.0) ? hit.normal : (hit.normal * -1.0);
        ray.o = hit.position + offsetNormal * 1e-3;
        ray.d = nextDir;
    }
    return out;
}

// pixelFootprintRadius, reference src/ProbeGather.cpp:740-793
RT_DEV double pixelFootprintRadius(const Camera& cam, int32_t px, int32_t py, const Hit& hit, double fallbackDistance)
{
    const int32_t nx = (px + 1 < cam.width) ? px + 1 : (px > 0) ? px - 1 : px;
    const Ray adjRay = buildPrimaryRay(cam, nx, py, nullptr);
    const double perpRadius = fallbackDistance * cam.tanPixelHalfAngle;
    const double denom = dot(adjRay.d, hit.normal);
    if (dabs(denom) > kEps)
    {
        const double t = dot(hit.position - adjRay.o, hit.normal) / denom;
        if (t > kEps)
        {
            const Vec adjHit = adjRay.o + adjRay.d * t;
            const double diffRadius = 0.5 * magnitude(adjHit - hit.position);
            if (diffRadius > 0.0)
            {
                return dmin(diffRadius, perpRadius);
            }
        }
    }
    return perpRadius;
}

// reflectedFootprintRadius, reference src/ProbeGather.cpp:818-853
RT_DEV double reflectedFootprintRadius(double tanPixelHalfAngle, double unfoldedPathLength, const Vec& viewer,
                                       const Hit& hit, bool haveAdjacent, const Vec& adjacentHit)
{
    const double viewerMag = magnitude(viewer);
    double perpFootprint = unfoldedPathLength * tanPixelHalfAngle;
    if (viewerMag > kEps)
    {
        const double cosView = dot(viewer / viewerMag, hit.normal);
        if (cosView > 0.0)
        {
            perpFootprint /= dmin(1.0, dmax(0.5, cosView));
        }
    }
    if (haveAdjacent)
    {
        const double diffRadius = 0.5 * magnitude(adjacentHit - hit.position);
        if (diffRadius > 0.0)
        {
            return dmin(diffRadius, perpFootprint);
        }
    }
    return perpFootprint;
}

// One pixel of ProbeGather::collectGatherPoints, reference src/ProbeGather.cpp:286-457.
// Calls emit(point) for every surviving sample, in the reference's order; returns the
// number of survivors.  stats[0..2] += camera rays, delta extensions, misses.
// DeferDelta (for pixels of one sample only): a pixel whose camera ray meets a delta surface is left
// alone — nothing emitted, nothing counted, -1 returned — for a second launch that follows the ray on
// (this instance then holds none of the code that does).
template <bool DeferDelta = false, class Emit>
RT_DEV int collectPixel(const SceneView& s, int32_t cameraIndex, int32_t px, int32_t py, unsigned long long* stats,
                        Emit&& emit)
{
    const Camera& staticCam = s.cameras[cameraIndex];
    Camera cam = staticCam;
    const bool motionActive = (s.shutterSpan > 0.0f) && (s.cameraTimeSamples > 1);
    const bool dofActive = (cam.projection == RT_PROJECTION_REALLENS) && (cam.apertureRadius > 0.0);
    const int motionSamples = motionActive ? s.cameraTimeSamples : 1;
    const int dofSamples = dofActive ? 16 : 1;
    const int primarySamples = dofSamples > motionSamples ? dofSamples : motionSamples;
    Rng rng;
    rng.tape = nullptr;
    rng.next = 0;
    rng.seed = s.probeSeed;
    rng.begin(0xFFFFFFFFu,
              (static_cast<uint64_t>(cameraIndex) << 40) | (static_cast<uint64_t>(py) * cam.width + px), 0);
    int survived = 0;
    for (int primary = 0; primary < primarySamples; ++primary)
    {
        const float sampleTime =
            motionActive ? s.frameTime + static_cast<float>(rng.value(static_cast<double>(s.shutterSpan))) : s.frameTime;
        // Camera::generatePrimaryRayAt, reference src/Camera.cpp:375-402: an animated camera is posed at the sample's time
        if (staticCam.animation >= 0)
        {
            cam = cameraAt(s, staticCam, sampleTime);
        }
        const Ray ray = buildPrimaryRay(cam, px, py, dofActive ? &rng : nullptr);
        stats[0] += 1;
        Hit first;
        if (!firstHit(s, ray, sampleTime, &first))
        {
            stats[2] += 1;
            continue;
        }
        bool direct = first.material < 0;
        if (!direct)
        {
            direct = !materialIsDelta(s.materials[first.material].type);
        }
        GatherPoint gp;
        gp.px = px;
        gp.py = py;
        gp.sampleTime = sampleTime;
        gp.sampleWeight = 1.0f;
        if (direct)
        {
            const Vec view = normalized(ray.o - first.position);
            gp.position[0] = first.position.x; gp.position[1] = first.position.y; gp.position[2] = first.position.z;
            gp.normal[0] = first.normal.x; gp.normal[1] = first.normal.y; gp.normal[2] = first.normal.z;
            gp.viewDir[0] = view.x; gp.viewDir[1] = view.y; gp.viewDir[2] = view.z;
            gp.material = first.material;
            gp.throughput[0] = 1.0f; gp.throughput[1] = 1.0f; gp.throughput[2] = 1.0f;
            gp.unfoldedPathLength = first.distance;
            gp.footprintRadius = pixelFootprintRadius(cam, px, py, first, first.distance);
            emit(gp, survived);
            survived += 1;
            continue;
        }
        if (DeferDelta)
        {
            stats[0] -= 1;
            return -1;
        }
        const bool stochasticDelta = s.materials[first.material].type == RT_MATERIAL_DIELECTRIC;
        const int extensionSamples = (stochasticDelta && !dofActive) ? 16 : 1;
        for (int ext = 0; ext < extensionSamples; ++ext)
        {
            const ExtendResult chain = extendAndRecord(s, rng, ray, sampleTime);
            if (chain.traversedDelta) stats[1] += 1;
            if (!chain.valid)
            {
                stats[2] += 1;
                continue;
            }
            const Vec reflectedView = neg(chain.finalDirection);
            Vec adjHit = mk(0, 0, 0);
            bool haveAdj = false;
            if (!stochasticDelta)
            {
                const int32_t adjX = (px + 1 < cam.width) ? px + 1 : (px > 0) ? px - 1 : px;
                if (adjX != px)
                {
                    const Ray adjRay = buildPrimaryRay(cam, adjX, py, dofActive ? &rng : nullptr);
                    Rng adjRng;
                    adjRng.tape = nullptr;
                    adjRng.next = 0;
                    adjRng.seed = 0xC0FFEEu;
                    adjRng.begin(0xFFFFFFFFu, 0, 0);
                    const ExtendResult adjChain = extendAndRecord(s, adjRng, adjRay, sampleTime);
                    if (adjChain.valid && adjChain.hit.material == chain.hit.material &&
                        dot(adjChain.hit.normal, chain.hit.normal) >= 0.5)
                    {
                        adjHit = adjChain.hit.position;
                        haveAdj = true;
                    }
                }
            }
            const Vec view = normalized(reflectedView);
            gp.position[0] = chain.hit.position.x; gp.position[1] = chain.hit.position.y; gp.position[2] = chain.hit.position.z;
            gp.normal[0] = chain.hit.normal.x; gp.normal[1] = chain.hit.normal.y; gp.normal[2] = chain.hit.normal.z;
            gp.viewDir[0] = view.x; gp.viewDir[1] = view.y; gp.viewDir[2] = view.z;
            gp.material = chain.hit.material;
            gp.throughput[0] = chain.throughput.r; gp.throughput[1] = chain.throughput.g; gp.throughput[2] = chain.throughput.b;
            gp.unfoldedPathLength = chain.unfoldedPathLength;
            gp.footprintRadius = reflectedFootprintRadius(cam.tanPixelHalfAngle, chain.unfoldedPathLength, reflectedView,
                                                          chain.hit, haveAdj, adjHit);
            emit(gp, survived);
            survived += 1;
        }
    }
    return survived;
}

}  // namespace rtd
