#include "rt_bvh.h"

#include <algorithm>
#include <limits>

namespace rtbvh
{

namespace
{

struct Builder
{
    const double* pos;
    std::vector<double> center;  // 3 per triangle: (a + b + c) / 3 per axis, reference src/Triangle.cpp:6-19
    FlatMesh out;

    double pivotOf(int32_t tri, int axis) const { return center[3 * tri + axis]; }

    void triangleBounds(int32_t tri, double* lo, double* hi) const
    {
        const double* p = pos + 9 * tri;
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = std::min({p[k], p[3 + k], p[6 + k]});
            hi[k] = std::max({p[k], p[3 + k], p[6 + k]});
        }
    }

    // reference src/Tree.cpp:18-36
    static int nextNonZeroAxis(const double* lo, const double* hi, int axis)
    {
        int valid = axis;
        while (hi[valid] - lo[valid] <= std::numeric_limits<double>::epsilon())
        {
            valid = (valid + 1) % 3;
            if (valid == axis)
            {
                break;
            }
        }
        return valid;
    }

    // reference src/Tree.cpp:172-205
    double axisMedian(const std::vector<int32_t>& objects, int axis) const
    {
        const size_t size = objects.size();
        if (size == 0) return 0.0;
        if (size == 1) return pivotOf(objects[0], axis);
        std::vector<double> pivots;
        pivots.reserve(size);
        for (int32_t t : objects)
        {
            pivots.push_back(pivotOf(t, axis));
        }
        std::sort(pivots.begin(), pivots.end());
        const size_t middle = size / 2;
        double median = pivots[middle];
        if (size % 2 == 0)
        {
            median += pivots[middle - 1];
            median /= 2.0;
        }
        return median;
    }

    // reference src/Tree.cpp:289-362; emits this node, then its left and right subtrees.
    void generate(const std::vector<int32_t>& objects, int axis, int depth)
    {
        const size_t self = out.nodes.size();
        out.nodes.push_back(FlatNode{});
        out.depth = std::max(out.depth, depth);
        FlatNode node{};
        node.triStart = static_cast<int32_t>(out.order.size());
        node.triCount = 0;
        node.right = -1;
        if (!objects.empty())
        {
            triangleBounds(objects[0], node.bmin, node.bmax);
            for (int32_t t : objects)
            {
                double lo[3], hi[3];
                triangleBounds(t, lo, hi);
                for (int k = 0; k < 3; ++k)
                {
                    node.bmin[k] = std::min(node.bmin[k], lo[k]);
                    node.bmax[k] = std::max(node.bmax[k], hi[k]);
                }
            }
            if (objects.size() <= 1)
            {
                out.order.push_back(objects[0]);
                node.triCount = 1;
            }
            else
            {
                const int splitAxis = nextNonZeroAxis(node.bmin, node.bmax, axis);
                const double pivot = axisMedian(objects, splitAxis);
                std::vector<int32_t> left, middle, right;
                for (int32_t t : objects)
                {
                    const double p = pivotOf(t, splitAxis);
                    if (p < pivot) left.push_back(t);
                    else if (p > pivot) right.push_back(t);
                    else middle.push_back(t);
                }
                for (int32_t t : middle)
                {
                    out.order.push_back(t);
                }
                node.triCount = static_cast<int32_t>(middle.size());
                if (!left.empty()) generate(left, (splitAxis + 1) % 3, depth + 1);
                if (!right.empty())
                {
                    node.right = static_cast<int32_t>(out.nodes.size());
                    generate(right, (splitAxis + 1) % 3, depth + 1);
                }
            }
        }
        node.skip = static_cast<int32_t>(out.nodes.size());
        out.nodes[self] = node;
    }
};

}  // namespace

FlatMesh build(const double* tri_pos, int64_t count)
{
    Builder b;
    b.pos = tri_pos;
    b.center.resize(static_cast<size_t>(3 * count));
    std::vector<int32_t> all(static_cast<size_t>(count));
    for (int64_t t = 0; t < count; ++t)
    {
        const double* p = tri_pos + 9 * t;
        for (int k = 0; k < 3; ++k)
        {
            b.center[3 * t + k] = (p[k] + p[3 + k] + p[6 + k]) / 3.0;
        }
        all[static_cast<size_t>(t)] = static_cast<int32_t>(t);
    }
    b.generate(all, 0, 1);
    return std::move(b.out);
}

}  // namespace rtbvh
