#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
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

// Binned surface-area build (16 bins per axis) over triangle indices; preorder emission.
struct SurfaceAreaBuilder
{
    const double* pos;
    double pad;
    int leafSize, maxDepth;
    std::vector<double> lo, hi, mid;  // 3 per triangle
    std::vector<int32_t> items;
    FlatMesh out;

    static double halfArea(const double* l, const double* h)
    {
        const double dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
        return dx * dy + dy * dz + dz * dx;
    }

    static int levelsFor(int64_t n, int leaf)
    {
        int levels = 0;
        int64_t capacity = leaf;
        while (capacity < n)
        {
            capacity *= 2;
            ++levels;
        }
        return levels;
    }

    void generate(size_t begin, size_t end, int depth)
    {
        const size_t self = out.nodes.size();
        out.nodes.push_back(FlatNode{});
        out.depth = std::max(out.depth, depth);
        FlatNode node{};
        node.triStart = static_cast<int32_t>(out.order.size());
        node.triCount = 0;
        node.right = -1;
        const size_t n = end - begin;
        double bl[3] = {0, 0, 0}, bh[3] = {0, 0, 0}, cl[3] = {0, 0, 0}, ch[3] = {0, 0, 0};
        for (size_t i = begin; i < end; ++i)
        {
            const int32_t t = items[i];
            for (int k = 0; k < 3; ++k)
            {
                if (i == begin)
                {
                    bl[k] = lo[3 * t + k];
                    bh[k] = hi[3 * t + k];
                    cl[k] = ch[k] = mid[3 * t + k];
                }
                else
                {
                    bl[k] = std::min(bl[k], lo[3 * t + k]);
                    bh[k] = std::max(bh[k], hi[3 * t + k]);
                    cl[k] = std::min(cl[k], mid[3 * t + k]);
                    ch[k] = std::max(ch[k], mid[3 * t + k]);
                }
            }
        }
        for (int k = 0; k < 3; ++k)
        {
            node.bmin[k] = bl[k] - pad;
            node.bmax[k] = bh[k] + pad;
        }
        if (n <= static_cast<size_t>(leafSize) || depth >= maxDepth)
        {
            for (size_t i = begin; i < end; ++i)
            {
                out.order.push_back(items[i]);
            }
            node.triCount = static_cast<int32_t>(n);
        }
        else
        {
            size_t middle = begin;
            // a balanced split is forced once the remaining levels are only just enough
            const bool forceBalanced = levelsFor(static_cast<int64_t>(n), leafSize) >= maxDepth - depth;
            if (!forceBalanced)
            {
                constexpr int kBins = 16;
                double bestCost = std::numeric_limits<double>::infinity();
                int bestAxis = -1, bestBin = 0;
                for (int axis = 0; axis < 3; ++axis)
                {
                    const double width = ch[axis] - cl[axis];
                    if (!(width > 0.0)) continue;
                    int64_t count[kBins] = {};
                    double binLo[kBins][3], binHi[kBins][3];
                    for (size_t i = begin; i < end; ++i)
                    {
                        const int32_t t = items[i];
                        int b = static_cast<int>(kBins * ((mid[3 * t + axis] - cl[axis]) / width));
                        b = std::min(kBins - 1, std::max(0, b));
                        for (int k = 0; k < 3; ++k)
                        {
                            binLo[b][k] = count[b] ? std::min(binLo[b][k], lo[3 * t + k]) : lo[3 * t + k];
                            binHi[b][k] = count[b] ? std::max(binHi[b][k], hi[3 * t + k]) : hi[3 * t + k];
                        }
                        ++count[b];
                    }
                    double rightArea[kBins] = {};
                    int64_t rightCount[kBins] = {};
                    double al[3], ah[3];
                    int64_t running = 0;
                    for (int b = kBins - 1; b > 0; --b)
                    {
                        if (count[b])
                        {
                            for (int k = 0; k < 3; ++k)
                            {
                                al[k] = running ? std::min(al[k], binLo[b][k]) : binLo[b][k];
                                ah[k] = running ? std::max(ah[k], binHi[b][k]) : binHi[b][k];
                            }
                            running += count[b];
                        }
                        rightCount[b] = running;
                        rightArea[b] = running ? halfArea(al, ah) : 0.0;
                    }
                    running = 0;
                    for (int b = 0; b < kBins - 1; ++b)  // split between bin b and b + 1
                    {
                        if (count[b])
                        {
                            for (int k = 0; k < 3; ++k)
                            {
                                al[k] = running ? std::min(al[k], binLo[b][k]) : binLo[b][k];
                                ah[k] = running ? std::max(ah[k], binHi[b][k]) : binHi[b][k];
                            }
                            running += count[b];
                        }
                        if (running == 0 || rightCount[b + 1] == 0) continue;
                        const double cost = halfArea(al, ah) * static_cast<double>(running) +
                                            rightArea[b + 1] * static_cast<double>(rightCount[b + 1]);
                        if (cost < bestCost)
                        {
                            bestCost = cost;
                            bestAxis = axis;
                            bestBin = b;
                        }
                    }
                }
                if (bestAxis >= 0)
                {
                    const double width = ch[bestAxis] - cl[bestAxis];
                    const auto split = std::partition(items.begin() + begin, items.begin() + end, [&](int32_t t) {
                        int b = static_cast<int>(kBins * ((mid[3 * t + bestAxis] - cl[bestAxis]) / width));
                        b = std::min(kBins - 1, std::max(0, b));
                        return b <= bestBin;
                    });
                    middle = static_cast<size_t>(split - items.begin());
                }
            }
            if (middle == begin || middle == end)
            {
                // equal counts along the widest axis of the centroids (ties broken by triangle index)
                int axis = 0;
                for (int k = 1; k < 3; ++k)
                {
                    if (ch[k] - cl[k] > ch[axis] - cl[axis]) axis = k;
                }
                middle = begin + n / 2;
                std::nth_element(items.begin() + begin, items.begin() + middle, items.begin() + end,
                                 [&](int32_t a, int32_t b) {
                                     const double ma = mid[3 * a + axis], mb = mid[3 * b + axis];
                                     return ma < mb || (ma == mb && a < b);
                                 });
            }
            generate(begin, middle, depth + 1);
            node.right = static_cast<int32_t>(out.nodes.size());
            generate(middle, end, depth + 1);
        }
        node.skip = static_cast<int32_t>(out.nodes.size());
        out.nodes[self] = node;
    }
};

}  // namespace

FlatMesh buildSurfaceArea(const double* tri_pos, int64_t count, const FlatMesh& reference, double pad, int leaf_size)
{
    SurfaceAreaBuilder b;
    b.pos = tri_pos;
    b.pad = pad;
    b.leafSize = std::max(1, leaf_size);
    // Depth is LDS: every level of inner nodes is one traversal-stack entry per lane.  Small meshes
    // get room for uneven splits (up to 17 levels in all, 16 of them inner); beyond that only two
    // levels above a balanced tree.
    const int balanced = 1 + SurfaceAreaBuilder::levelsFor(count, b.leafSize);
    b.maxDepth = std::min(44, std::max(balanced + 2, std::min(balanced + 6, 17)));
    b.lo.resize(static_cast<size_t>(3 * count));
    b.hi.resize(static_cast<size_t>(3 * count));
    b.mid.resize(static_cast<size_t>(3 * count));
    b.items.resize(static_cast<size_t>(count));
    for (int64_t t = 0; t < count; ++t)
    {
        const double* p = tri_pos + 9 * t;
        for (int k = 0; k < 3; ++k)
        {
            b.lo[3 * t + k] = std::min({p[k], p[3 + k], p[6 + k]});
            b.hi[3 * t + k] = std::max({p[k], p[3 + k], p[6 + k]});
            b.mid[3 * t + k] = 0.5 * (b.lo[3 * t + k] + b.hi[3 * t + k]);
        }
        b.items[static_cast<size_t>(t)] = static_cast<int32_t>(t);
    }
    b.generate(0, static_cast<size_t>(count), 1);

    // per load-order triangle: its rank in the reference's visiting order and its reference node
    std::vector<int32_t> rankOf(static_cast<size_t>(count), 0), nodeOf(static_cast<size_t>(count), 0);
    for (size_t n = 0; n < reference.nodes.size(); ++n)
    {
        const FlatNode& node = reference.nodes[n];
        for (int32_t slot = node.triStart; slot < node.triStart + node.triCount; ++slot)
        {
            rankOf[static_cast<size_t>(reference.order[static_cast<size_t>(slot)])] = slot;
            nodeOf[static_cast<size_t>(reference.order[static_cast<size_t>(slot)])] = static_cast<int32_t>(n);
        }
    }
    FlatMesh out = std::move(b.out);
    out.rank.reserve(out.order.size());
    out.referenceBox.reserve(6 * out.order.size());
    for (int32_t original : out.order)
    {
        out.rank.push_back(rankOf[static_cast<size_t>(original)]);
        const FlatNode& node = reference.nodes[static_cast<size_t>(nodeOf[static_cast<size_t>(original)])];
        out.referenceBox.insert(out.referenceBox.end(), node.bmin, node.bmin + 3);
        out.referenceBox.insert(out.referenceBox.end(), node.bmax, node.bmax + 3);
    }
    return out;
}

namespace
{

float roundDown(double v)
{
    float f = static_cast<float>(v);
    if (static_cast<double>(f) > v) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}

float roundUp(double v)
{
    float f = static_cast<float>(v);
    if (static_cast<double>(f) < v) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

}  // namespace

WideMesh widen(const FlatMesh& tree)
{
    WideMesh out;
    const size_t count = tree.nodes.size();
    auto isLeaf = [&](size_t i) { return tree.nodes[i].skip == static_cast<int32_t>(i) + 1; };
    // inner nodes keep their preorder
    std::vector<int32_t> wideIndex(count, -1);
    int32_t inner = 0;
    for (size_t i = 0; i < count; ++i)
    {
        if (!isLeaf(i)) wideIndex[i] = inner++;
    }
    auto reference = [&](size_t i) {
        const FlatNode& node = tree.nodes[i];
        if (!isLeaf(i)) return wideIndex[i];
        return node.triCount > 0 ? leafReference(node.triStart, node.triCount) : kNoChild;
    };
    auto setBox = [&](float* box, const FlatNode& node) {
        for (int k = 0; k < 3; ++k)
        {
            box[k] = roundDown(node.bmin[k]);
            box[3 + k] = roundUp(node.bmax[k]);
        }
    };
    if (count == 0 || isLeaf(0))
    {
        // a mesh that fits one leaf (or has no triangles): a root whose only child is that leaf
        WideNode root{};
        root.left = root.right = kNoChild;
        if (count > 0)
        {
            root.left = reference(0);
            setBox(root.leftBox, tree.nodes[0]);
        }
        out.nodes.push_back(root);
        out.pushDepth = 0;
        return out;
    }
    out.nodes.resize(static_cast<size_t>(inner));
    // depth of inner nodes along the deepest path = entries the stack may hold
    std::vector<int> level(count, 0);
    for (size_t i = 0; i < count; ++i)
    {
        if (isLeaf(i)) continue;
        const FlatNode& node = tree.nodes[i];
        const size_t left = i + 1, right = static_cast<size_t>(node.right);
        WideNode wide{};
        wide.left = reference(left);
        wide.right = reference(right);
        setBox(wide.leftBox, tree.nodes[left]);
        setBox(wide.rightBox, tree.nodes[right]);
        out.nodes[static_cast<size_t>(wideIndex[i])] = wide;
        level[left] = level[right] = level[i] + 1;
        out.pushDepth = std::max(out.pushDepth, level[i] + 1);
    }
    return out;
}

WideMesh breadthFirst(const WideMesh& mesh)
{
    WideMesh out;
    out.pushDepth = mesh.pushDepth;
    const size_t count = mesh.nodes.size();
    std::vector<int32_t> newIndex(count, -1), order;
    order.reserve(count);
    if (count > 0)
    {
        newIndex[0] = 0;
        order.push_back(0);
    }
    for (size_t head = 0; head < order.size(); ++head)
    {
        const WideNode& node = mesh.nodes[static_cast<size_t>(order[head])];
        for (int32_t child : {node.left, node.right})
        {
            if (child >= 0 && newIndex[static_cast<size_t>(child)] < 0)
            {
                newIndex[static_cast<size_t>(child)] = static_cast<int32_t>(order.size());
                order.push_back(child);
            }
        }
    }
    out.nodes.reserve(order.size());
    for (int32_t old : order)
    {
        WideNode node = mesh.nodes[static_cast<size_t>(old)];
        if (node.left >= 0) node.left = newIndex[static_cast<size_t>(node.left)];
        if (node.right >= 0) node.right = newIndex[static_cast<size_t>(node.right)];
        out.nodes.push_back(node);
    }
    return out;
}

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
    FlatMesh out = std::move(b.out);
    // every triangle's rank is its slot; its box is the box of the node whose page holds it
    out.rank.resize(out.order.size());
    out.referenceBox.resize(6 * out.order.size());
    for (const FlatNode& node : out.nodes)
    {
        for (int32_t slot = node.triStart; slot < node.triStart + node.triCount; ++slot)
        {
            out.rank[static_cast<size_t>(slot)] = slot;
            std::copy(node.bmin, node.bmin + 3, out.referenceBox.begin() + 6 * slot);
            std::copy(node.bmax, node.bmax + 3, out.referenceBox.begin() + 6 * slot + 3);
        }
    }
    return out;
}

}  // namespace rtbvh
