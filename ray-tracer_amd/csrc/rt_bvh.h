// Host-side tree build + flattening for one mesh.
//
// The tree is the reference's median-split structure with page size 1
// (reference src/Tree.cpp:18-36,172-205,289-362; include/Mesh.h:15-19): nodes are
// emitted in the order Tree::castRayIntoNode visits them (page, left subtree, right
// subtree), each with the index of the node that follows its subtree and of its
// right child.  Triangles are re-ordered so that every node's page is a contiguous
// run; a triangle's position in that order is its rank in the reference's visiting
// order, which is what breaks ties between equally distant hits.
#pragma once

#include <climits>
#include <cstdint>
#include <vector>

namespace rtbvh
{

struct FlatNode
{
    double bmin[3];
    double bmax[3];
    int32_t skip;
    int32_t triStart;
    int32_t triCount;
    int32_t right;  // index of the right child, -1 when there is none; a left child, when present, is node + 1
};

struct FlatMesh
{
    std::vector<FlatNode> nodes;
    std::vector<int32_t> order;  // slot -> triangle index in load order
    int depth = 0;
    // Per slot: the triangle's rank in the reference's visiting order and the box (min xyz,
    // max xyz) of the reference node whose page holds it.
    std::vector<int32_t> rank;
    std::vector<double> referenceBox;
};

// The reference's tree.  tri_pos: 9 doubles per triangle (a, b, c), `count` triangles in load order.
FlatMesh build(const double* tri_pos, int64_t count);

// A tree of better quality over the same triangles: binned surface-area splits, up to
// `leaf_size` triangles per leaf, none in inner nodes, every box grown by `pad` on all sides.
// The closest hit the reference reports is "of the triangles whose OWN reference node's box
// passes the slab test and whose watertight test hits, the nearest, earliest-visited first":
// a reference node's box lies inside every ancestor's (each is the exact min / max over a
// superset of triangles) and the slab test's rounded operations are monotone in the box
// planes, so passing the own node's box implies passing every ancestor's.  The tree
// therefore only has to REACH every such triangle (hence the padding: a ray that hits a
// triangle meets the padded box that holds it whatever the rounding); `rank` and
// `referenceBox` let the walk apply the reference's admission test and tie-break per hit.
FlatMesh buildSurfaceArea(const double* tri_pos, int64_t count, const FlatMesh& reference, double pad, int leaf_size);

// The surface-area tree as the device walks it: inner nodes only, each holding BOTH children's
// boxes (single precision, rounded outward — the boxes only have to contain, see above) and a
// reference per child: >= 0 the index of an inner node, kNoChild, or a leaf as
// leafReference(first slot, count).  One 64-byte node per walk step, nothing to fetch for a leaf.
struct WideNode
{
    float leftBox[6];   // min xyz, max xyz
    float rightBox[6];
    int32_t left, right;
    int32_t reserved[2];
};
static_assert(sizeof(WideNode) == 64, "one node per 64-byte line");

constexpr int32_t kNoChild = INT32_MIN;
constexpr int kLeafCountBits = 4;  // up to 16 triangles per leaf
inline int32_t leafReference(int32_t first, int32_t count) { return -1 - ((first << kLeafCountBits) | (count - 1)); }

struct WideMesh
{
    std::vector<WideNode> nodes;  // node 0 is the root (present even for an empty mesh)
    int pushDepth = 0;            // most entries the walk's stack can hold at once
};

WideMesh widen(const FlatMesh& tree);

// The same nodes numbered breadth-first from the root (widen() numbers them in preorder): the
// first k nodes are then the top of the tree, the part every ray walks through, which a kernel
// can keep in LDS as one contiguous block.
WideMesh breadthFirst(const WideMesh& mesh);

}  // namespace rtbvh
