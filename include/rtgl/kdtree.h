// kdtree.h -- AABB / KdNode layouts of the reference (src/kdtree.h:26-30,63-69; KdNode mirrors the
// shader's `Node`, shaders/raytracer.glsl:29-36, 48 bytes) and a KdTree builder with the reference's
// construction rule, restated: leaves at size <= NODE_SIZE or depth >= MAX_DEPTH; otherwise sort by the
// lower bound on axis depth%3, split at the median element's lower bound, put every primitive into each
// side its box overlaps (straddlers are duplicated), recurse left then right; node ids in pre-order.
// The shader never culls by these boxes (its AABB test is compiled out, :288-292); only the node order,
// offsets and counts matter for rendering.
#pragma once
#include <algorithm>
#include <limits>
#include <ostream>
#include <vector>

#include "glm_shim.h"

using uint = unsigned int;
constexpr uint INVALID = std::numeric_limits<uint>::max();

struct Ray { glm::vec3 origin; glm::vec3 direction; };
struct AABB { glm::vec4 min; glm::vec4 max; };

inline bool intersect(const AABB *a, const AABB *b)
{
    for (int k = 0; k < 3; ++k)
        if (!(a->min[k] <= b->max[k] && a->max[k] >= b->min[k])) return false;
    return true;
}

// stream form of the reference (src/kdtree.h:56-60); its main() prints nodes this way (src/main.cpp:165)
inline std::ostream &operator<<(std::ostream &os, const AABB &obj)
{
    os << "AABB { min = " << obj.min << ", max = " << obj.max << " }";
    return os;
}

struct KdNode : public AABB
{
    uint left = INVALID;
    uint right = INVALID;
    uint offset = 0;
    uint count = 0;
};

// src/kdtree.h:71-75
inline std::ostream &operator<<(std::ostream &os, const KdNode &obj)
{
    os << "Node { l = " << obj.left << ", r = " << obj.right << ", o = " << obj.offset << ", c = " << obj.count << " }";
    return os;
}
static_assert(sizeof(KdNode) == 48, "KdNode must match the shader's Node (std430, 48 bytes)");

template <class Bounded, uint NODE_SIZE = 8, uint MAX_DEPTH = 5>
class KdTree
{
public:
    explicit KdTree(const std::vector<Bounded> &primitives) { build(primitives, bounds(primitives), 0); }

    static AABB bounds(const std::vector<Bounded> &primitives)
    {
        AABB total{glm::vec4(+1e5f), glm::vec4(-1e5f)};
        for (const Bounded &p : primitives) {
            AABB b = p.bounds();
            total.min = glm::min(b.min, total.min);
            total.max = glm::max(b.max, total.max);
        }
        return total;
    }

    std::vector<KdNode> nodes() const { return m_nodes; }
    std::vector<Bounded> primitives() const { return m_primitives; }

private:
    uint build(std::vector<Bounded> prims, const AABB &box, uint depth)
    {
        const uint id = (uint)m_nodes.size();
        KdNode node;
        node.min = box.min; node.max = box.max;
        if (prims.size() <= NODE_SIZE || depth >= MAX_DEPTH) {
            node.offset = (uint)m_primitives.size();
            node.count = (uint)prims.size();
            m_primitives.insert(m_primitives.end(), prims.begin(), prims.end());
            m_nodes.push_back(node);
            return id;
        }
        m_nodes.push_back(KdNode{});
        const int axis = (int)(depth % 3);
        std::sort(prims.begin(), prims.end(), [axis](const Bounded &a, const Bounded &b) { return a.bounds().min[axis] < b.bounds().min[axis]; });
        const float boundary = prims[prims.size() / 2].bounds().min[axis];
        AABB lbox = box, rbox = box;
        lbox.max[axis] = boundary - 0.001f;
        rbox.min[axis] = boundary;
        std::vector<Bounded> left, right;
        for (const Bounded &p : prims) {
            AABB b = p.bounds();
            const bool in_l = intersect(&lbox, &b), in_r = intersect(&rbox, &b);
            if (in_l) left.push_back(p);
            if (in_r) right.push_back(p);
        }
        node.left = left.empty() ? INVALID : build(left, lbox, depth + 1);
        node.right = right.empty() ? INVALID : build(right, rbox, depth + 1);
        m_nodes[id] = node;
        return id;
    }

    std::vector<KdNode> m_nodes;
    std::vector<Bounded> m_primitives;
};
