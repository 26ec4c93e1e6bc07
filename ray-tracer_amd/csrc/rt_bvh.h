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
    std::vector<int32_t> order;  // page-ordered slot -> triangle index in load order
    int depth = 0;
};

// tri_pos: 9 doubles per triangle (a, b, c), `count` triangles in load order.
FlatMesh build(const double* tri_pos, int64_t count);

}  // namespace rtbvh
