// Host-side builder of the alternative accelerator: a 4-wide bounding volume hierarchy over the scene's
// preprocessed triangles, built top-down with the binned surface area heuristic.  The reference has one
// accelerator only, BSPTree (src/bsp.hpp:35-48); this one stands behind the same nearest-intersection
// query (src/bsp.hpp:50-51) as an option: every triangle is stored once, leaves hold at most a handful of
// triangles, and a node with its four child boxes is one 64-byte record.
//
// The build is deterministic: the tree depends only on the triangles and the parameters, not on the number
// of threads (subtrees are built in parallel, node numbers are assigned afterwards in one sequential pass).
#include "rt_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

int rt_internal_fail(int code, const char* message);

namespace {

constexpr int kBins = 64;
constexpr uint32_t kMaxDepth2 = 48;         // binary levels before splits fall back to medians
constexpr uint32_t kParallelGrain = 1u << 15;

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -std::numeric_limits<float>::infinity(); }
    }
    void grow(const Box& b) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    void grow(const float* p) {
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); }
    }
    // half the surface area; 0 for an empty box
    double area() const {
        double const dx = double(hi[0]) - lo[0], dy = double(hi[1]) - lo[1], dz = double(hi[2]) - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Prim { Box box; uint32_t id; };

// Binary node: count > 0 marks a leaf over prims[first, first+count).
struct Node2 { Box box; uint32_t left = 0, right = 0, first = 0, count = 0; };

inline float centroid(const Prim& p, int axis) { return 0.5f * p.box.lo[axis] + 0.5f * p.box.hi[axis]; }

struct Builder {
    std::vector<Prim> prims;
    std::vector<Node2> nodes;
    std::atomic<uint32_t> nextNode{1};
    uint32_t maxLeaf = 4;
    double nodeCost = 1.0;          // cost of one more node visit, in triangle tests

    // work queue of subtrees still to split
    struct Task { uint32_t node, first, count, depth; };
    std::mutex mutex;
    std::condition_variable wake;
    std::vector<Task> queue;
    uint32_t pending = 0;
    bool failed = false;

    // Splits prims[first, first+count) of `node`; returns false when the node became a leaf.
    bool split(uint32_t node, uint32_t first, uint32_t count, uint32_t depth, uint32_t& mid) {
        Prim* const P = prims.data() + first;
        Box bounds, cb;
        bounds.reset(); cb.reset();
        for (uint32_t i = 0; i < count; ++i) {
            bounds.grow(P[i].box);
            float const c[3] = {centroid(P[i], 0), centroid(P[i], 1), centroid(P[i], 2)};
            cb.grow(c);
        }
        nodes[node].box = bounds;
        auto makeLeaf = [&] { nodes[node].first = first; nodes[node].count = count; return false; };
        if (count == 1) return makeLeaf();

        // Levels left before the depth limit: once only medians can still finish in time, use medians.
        uint32_t levelsNeeded = 0;
        for (uint64_t cap = maxLeaf; cap < count; cap *= 2) ++levelsNeeded;
        bool const forceMedian = depth + levelsNeeded + 1 >= kMaxDepth2;

        int bestAxis = -1, bestBin = -1;
        double bestCost = std::numeric_limits<double>::infinity();
        if (!forceMedian) {
            for (int axis = 0; axis < 3; ++axis) {
                float const lo = cb.lo[axis], extent = cb.hi[axis] - cb.lo[axis];
                if (!(extent > 0.0f)) continue;
                float const scale = float(kBins) / extent;
                Box binBox[kBins];
                uint32_t binCount[kBins] = {};
                for (auto& b : binBox) b.reset();
                for (uint32_t i = 0; i < count; ++i) {
                    int const b = std::min(kBins - 1, std::max(0, int((centroid(P[i], axis) - lo) * scale)));
                    binBox[b].grow(P[i].box);
                    ++binCount[b];
                }
                double rightArea[kBins];
                Box acc;
                acc.reset();
                for (int b = kBins - 1; b > 0; --b) { acc.grow(binBox[b]); rightArea[b] = acc.area(); }
                acc.reset();
                uint32_t leftCount = 0;
                for (int b = 0; b + 1 < kBins; ++b) {
                    acc.grow(binBox[b]);
                    leftCount += binCount[b];
                    if (leftCount == 0 || leftCount == count) continue;
                    double const cost = acc.area() * leftCount + rightArea[b + 1] * (count - leftCount);
                    if (cost < bestCost) { bestCost = cost; bestAxis = axis; bestBin = b; }
                }
            }
        }
        if (count <= maxLeaf) {
            // surface area heuristic with unit costs: a leaf of n triangles against one more node visit
            double const area = bounds.area();
            double const splitCost = (bestAxis >= 0 && area > 0.0) ? nodeCost + bestCost / area : std::numeric_limits<double>::infinity();
            if (double(count) <= splitCost) return makeLeaf();
        }
        if (bestAxis >= 0) {
            float const lo = cb.lo[bestAxis], scale = float(kBins) / (cb.hi[bestAxis] - cb.lo[bestAxis]);
            Prim* const m = std::partition(P, P + count, [&](const Prim& p) {
                return std::min(kBins - 1, std::max(0, int((centroid(p, bestAxis) - lo) * scale))) <= bestBin;
            });
            mid = uint32_t(m - P);
        }
        else {
            // no usable plane (coincident centroids) or out of levels: halve along the widest centroid axis
            int axis = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
            mid = count / 2;
            std::nth_element(P, P + mid, P + count, [&](const Prim& a, const Prim& b) {
                float const ca = centroid(a, axis), cbv = centroid(b, axis);
                return ca < cbv || (ca == cbv && a.id < b.id);
            });
        }
        if (mid == 0 || mid == count) mid = count / 2;
        return true;
    }

    void subtree(uint32_t node, uint32_t first, uint32_t count, uint32_t depth) {
        uint32_t mid = 0;
        if (!split(node, first, count, depth, mid)) return;
        uint32_t const pair = nextNode.fetch_add(2);
        nodes[node].left = pair;
        nodes[node].right = pair + 1;
        subtree(pair, first, mid, depth + 1);
        subtree(pair + 1, first + mid, count - mid, depth + 1);
    }

    void worker() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lock(mutex);
                wake.wait(lock, [&] { return !queue.empty() || pending == 0 || failed; });
                if (failed || (queue.empty() && pending == 0)) return;
                t = queue.back();
                queue.pop_back();
            }
            try {
                if (t.count <= kParallelGrain) subtree(t.node, t.first, t.count, t.depth);
                else {
                    uint32_t mid = 0;
                    if (split(t.node, t.first, t.count, t.depth, mid)) {
                        uint32_t const pair = nextNode.fetch_add(2);
                        nodes[t.node].left = pair;
                        nodes[t.node].right = pair + 1;
                        std::lock_guard<std::mutex> lock(mutex);
                        queue.push_back({pair, t.first, mid, t.depth + 1});
                        queue.push_back({pair + 1, t.first + mid, t.count - mid, t.depth + 1});
                        pending += 2;
                    }
                }
            }
            catch (...) {
                std::lock_guard<std::mutex> lock(mutex);
                failed = true;
            }
            {
                std::lock_guard<std::mutex> lock(mutex);
                --pending;
            }
            wake.notify_all();
        }
    }
};

}  // namespace

namespace {

// Stores up to four child boxes in a node as 8-bit offsets from the node's lower corner.  Per axis the step is
// the smallest power of two with 255 steps spanning the children; a lower plane is rounded down and an upper
// plane up, and each stored plane is checked as the traversal decodes it (RN(origin + q * step)), so the stored
// box always contains the box it was made from.
void quantise(rt_bvh4_node& node, const Box* boxes, int count) {
    Box all;
    all.reset();
    for (int k = 0; k < count; ++k) all.grow(boxes[k]);
    uint8_t* const qlo[3] = {node.qlo_x, node.qlo_y, node.qlo_z};
    uint8_t* const qhi[3] = {node.qhi_x, node.qhi_y, node.qhi_z};
    for (int a = 0; a < 3; ++a) {
        float const origin = all.lo[a];
        node.origin[a] = origin;
        double const extent = double(all.hi[a]) - double(origin);
        int e = 1;                                           // biased exponent of the step
        if (extent > 0.0) {
            int ex = 0;
            (void)std::frexp(extent / 255.0, &ex);           // extent/255 = m * 2^ex, m in [0.5, 1)
            e = std::min(254, std::max(1, ex + 127));        // step = 2^ex >= extent/255
        }
        for (;;) {
            double const step = std::ldexp(1.0, e - 127);
            bool fits = true;
            for (int k = 0; k < count && fits; ++k) {
                double lo = std::floor((double(boxes[k].lo[a]) - origin) / step);
                double hi = std::ceil((double(boxes[k].hi[a]) - origin) / step);
                lo = std::max(0.0, lo);
                // as decoded by the traversal: float(q) * step is exact, the sum is rounded once
                while (lo > 0.0 && float(double(origin) + lo * step) > boxes[k].lo[a]) lo -= 1.0;
                while (hi <= 255.0 && float(double(origin) + hi * step) < boxes[k].hi[a]) hi += 1.0;
                if (hi > 255.0 || float(double(origin) + lo * step) > boxes[k].lo[a]) { fits = false; break; }
                qlo[a][k] = uint8_t(lo);
                qhi[a][k] = uint8_t(hi);
            }
            if (fits || e >= 254) break;
            ++e;
        }
        node.exp[a] = uint8_t(e);
    }
}

}  // namespace

struct rt_bvh4 {
    std::vector<rt_bvh4_node> nodes;
    std::vector<uint32_t> order;
    uint32_t depth = 0, stackNeed = 0, leafCount = 0, maxLeafFound = 0;
};

extern "C" {

int rt_bvh4_build(const float* pretris, uint32_t pretri_count, uint32_t max_leaf_tris, int threads, rt_bvh4** out) {
    if (!out || (pretri_count && !pretris)) return rt_internal_fail(RT_ERR_INVALID, "rt_bvh4_build: null argument");
    *out = nullptr;
    if (max_leaf_tris == 0 || max_leaf_tris > 8) return rt_internal_fail(RT_ERR_INVALID, "rt_bvh4_build: max_leaf_tris must be 1..8");
    if (pretri_count >= (1u << 28) - 1u) return rt_internal_fail(RT_ERR_LIMIT, "rt_bvh4_build: more than 2^28 triangles");
    try {
        Builder B;
        B.maxLeaf = max_leaf_tris;
        if (const char* text = std::getenv("RT_AMD_BVH_NODE_COST")) B.nodeCost = std::max(0.0, std::atof(text));      // tuning aid
        B.prims.resize(pretri_count);
        float absMax = 0.0f;
        for (uint32_t i = 0; i < pretri_count; ++i) {
            // {normal, v1, v1ToV3, v1ToV2}: the corners as the intersection test sees them
            float const* t = pretris + 12 * size_t(i);
            float const a[3] = {t[3], t[4], t[5]};
            float const c[3] = {t[3] + t[6], t[4] + t[7], t[5] + t[8]};
            float const b[3] = {t[3] + t[9], t[4] + t[10], t[5] + t[11]};
            Prim& p = B.prims[i];
            p.box.reset();
            p.box.grow(a); p.box.grow(b); p.box.grow(c);
            p.id = i;
            for (int k = 0; k < 3; ++k) {
                if (!(std::fabs(p.box.lo[k]) < std::numeric_limits<float>::infinity()) || !(std::fabs(p.box.hi[k]) < std::numeric_limits<float>::infinity()))
                    return rt_internal_fail(RT_ERR_INVALID, "rt_bvh4_build: triangle with a non-finite corner");
                absMax = std::max(absMax, std::max(std::fabs(p.box.lo[k]), std::fabs(p.box.hi[k])));
            }
        }
        std::unique_ptr<rt_bvh4> R(new rt_bvh4);
        R->order.resize(pretri_count);
        rt_bvh4_node empty;
        std::memset(&empty, 0, sizeof empty);
        for (int k = 0; k < 4; ++k) empty.child[k] = 0xFFFFFFFFu;
        if (pretri_count == 0) {
            R->nodes.push_back(empty);
            *out = R.release();
            return RT_OK;
        }
        B.nodes.resize(2 * size_t(pretri_count) + 2);
        unsigned workers = threads > 0 ? unsigned(threads) : 0u;
        if (workers == 0) {
            if (const char* text = std::getenv("OMP_NUM_THREADS")) workers = unsigned(std::max(1, std::atoi(text)));
            else workers = std::min(std::thread::hardware_concurrency(), 16u);
        }
        workers = std::max(1u, std::min(workers, 64u));
        B.queue.push_back({0u, 0u, pretri_count, 0u});
        B.pending = 1;
        if (workers == 1 || pretri_count <= kParallelGrain) B.worker();
        else {
            std::vector<std::thread> pool;
            for (unsigned w = 0; w < workers; ++w) pool.emplace_back([&B] { B.worker(); });
            for (auto& t : pool) t.join();
        }
        if (B.failed) return rt_internal_fail(RT_ERR_NOMEM, "rt_bvh4_build: out of memory in a worker");

        // ---- collapse to four children per node, number the nodes, pad the boxes ------------------------------
        // A node's interior children get consecutive numbers (they are the lines a traversal step may touch next);
        // subtrees follow depth-first.  Child boxes are padded by 2^-18 of the largest coordinate, which covers the
        // rounding of the slab test and of the corner sums above, and then quantised outwards.
        float const pad = absMax * 0x1p-18f;
        struct Item { uint32_t node2, node4, depth; };
        std::vector<Item> todo;
        R->nodes.push_back(empty);
        todo.push_back({0u, 0u, 1u});
        auto leafRef = [&](const Node2& n) { return 0x80000000u | ((n.count - 1u) << 28) | n.first; };
        auto padded = [&](const Box& b) {
            Box p = b;
            for (int k = 0; k < 3; ++k) { p.lo[k] -= pad; p.hi[k] += pad; }
            return p;
        };
        if (B.nodes[0].count > 0) {          // the whole scene is one leaf
            rt_bvh4_node& n4 = R->nodes[0];
            const Node2& n = B.nodes[0];
            Box const only = padded(n.box);
            quantise(n4, &only, 1);
            n4.child[0] = leafRef(n);
            R->leafCount = 1;
            R->maxLeafFound = n.count;
            R->depth = 1;
            todo.clear();
        }
        while (!todo.empty()) {
            Item const it = todo.back();
            todo.pop_back();
            R->depth = std::max(R->depth, it.depth);
            uint32_t kids[4] = {B.nodes[it.node2].left, B.nodes[it.node2].right, 0, 0};
            int n = 2;
            while (n < 4) {
                int pick = -1;
                double best = -1.0;
                for (int k = 0; k < n; ++k) {
                    const Node2& c = B.nodes[kids[k]];
                    if (c.count == 0 && c.box.area() > best) { best = c.box.area(); pick = k; }
                }
                if (pick < 0) break;
                uint32_t const opened = kids[pick];
                for (int k = n; k > pick + 1; --k) kids[k] = kids[k - 1];
                kids[pick] = B.nodes[opened].left;
                kids[pick + 1] = B.nodes[opened].right;
                ++n;
            }
            uint32_t firstInterior = uint32_t(R->nodes.size());
            int interior = 0;
            for (int k = 0; k < n; ++k) if (B.nodes[kids[k]].count == 0) ++interior;
            R->nodes.resize(R->nodes.size() + size_t(interior), empty);
            rt_bvh4_node& n4 = R->nodes[it.node4];
            uint32_t next = firstInterior;
            Item pushed[4];
            int pushedCount = 0;
            Box childBoxes[4];
            for (int k = 0; k < n; ++k) childBoxes[k] = padded(B.nodes[kids[k]].box);
            quantise(n4, childBoxes, n);
            for (int k = 0; k < n; ++k) {
                const Node2& c = B.nodes[kids[k]];
                if (c.count > 0) {
                    n4.child[k] = leafRef(c);
                    ++R->leafCount;
                    R->maxLeafFound = std::max(R->maxLeafFound, c.count);
                }
                else {
                    n4.child[k] = next;
                    pushed[pushedCount++] = {kids[k], next, it.depth + 1};
                    ++next;
                }
            }
            for (int k = pushedCount - 1; k >= 0; --k) todo.push_back(pushed[k]);
        }
        for (uint32_t i = 0; i < pretri_count; ++i) R->order[i] = B.prims[i].id;
        // Largest number of deferred children a traversal can hold: children have larger numbers than their
        // parent, so one backward pass suffices.
        std::vector<uint32_t> need(R->nodes.size(), 0u);
        for (size_t i = R->nodes.size(); i-- > 0;) {
            const rt_bvh4_node& n4 = R->nodes[i];
            uint32_t present = 0, deepest = 0;
            for (int k = 0; k < 4; ++k) {
                if (n4.child[k] == 0xFFFFFFFFu) continue;
                ++present;
                if ((n4.child[k] & 0x80000000u) == 0u) deepest = std::max(deepest, need[n4.child[k]]);
            }
            need[i] = (present ? present - 1u : 0u) + deepest;
        }
        R->stackNeed = need[0];
        *out = R.release();
        return RT_OK;
    }
    catch (std::bad_alloc const&) { return rt_internal_fail(RT_ERR_NOMEM, "rt_bvh4_build: out of host memory"); }
    catch (...) { return rt_internal_fail(RT_ERR_INVALID, "rt_bvh4_build: unexpected exception"); }
}

void rt_bvh4_destroy(rt_bvh4* bvh) { delete bvh; }

void rt_bvh4_counts(const rt_bvh4* bvh, uint32_t counts[6]) {
    if (!bvh || !counts) return;
    counts[0] = uint32_t(bvh->nodes.size());
    counts[1] = uint32_t(bvh->order.size());
    counts[2] = bvh->depth;
    counts[3] = bvh->stackNeed;
    counts[4] = bvh->leafCount;
    counts[5] = bvh->maxLeafFound;
}

void rt_bvh4_view(const rt_bvh4* bvh, const rt_bvh4_node** nodes, const uint32_t** tri_order) {
    if (!bvh) return;
    if (nodes) *nodes = bvh->nodes.data();
    if (tri_order) *tri_order = bvh->order.data();
}

}  // extern "C"
