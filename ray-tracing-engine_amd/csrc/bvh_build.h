// bvh_build.h — host-side builder of the flattened scene acceleration data.
//
// The reference's BVH (source/BVH.h, source/AABB.cpp) is dead, unlinked and has
// different semantics from the brute-force loop that defines the result
// (SURVEY.md §0.2, App. A.4), so nothing of it is reproduced.  This is our own
// design for gfx950:
//   * ONE binary BVH over all meshes (not one pointer tree per mesh);
//   * node records holding BOTH child boxes, so a lane fetches one record per step and
//     never touches a child it culls — built as 64-byte float records (Node: tests,
//     export) and traversed as 32-byte binary16-packed records (Node16, see below);
//   * 48-byte triangle records (p0, e1, e2, global id, mesh) stored in leaf
//     order, so a leaf is 1..leaf_max consecutive records;
//   * boxes padded by an absolute epsilon and a depth bound, so the LDS
//     traversal stack (csrc/rt_kernels.hip) has a fixed size.
// Exactness: traversal must return what RayTracer::rayTrace (reference
// source/RayTracer.h:27-53) returns — the closest positive t, lowest
// (mesh, triangle) index on ties.  Any BVH does as long as no box ever culls a
// triangle the float Möller–Trumbore test accepts; padding + a conservative
// slab test guarantee that (DESIGN.md §BVH exactness).
#pragma once

#include <cstdint>
#include <functional>
#include <vector>

#include "rt_amd.h"

namespace rtbvh {

struct alignas(16) Node {   // 64 B, full-precision form (host build, tests)
  float lo0[3], hi0[3];     // box of child 0
  float lo1[3], hi1[3];     // box of child 1
  int32_t child[2];         // >= 0: node index; < 0: leaf, ~child = first << 3 | (count - 1)
  uint32_t pad[2];
};
static_assert(sizeof(Node) == 64, "node record must be 64 bytes");

// What the GPU traverses: the same node with its 12 box planes stored as IEEE
// binary16 of (coordinate * boxScale), lower planes rounded DOWN and upper planes
// rounded UP, so a packed box always contains the float box.  Halves the bytes a
// lane pulls through the CU's 64 B/clk L1 data path per traversal step — the unit
// that bounds this kernel on cache-resident scenes.  boxScale is a power of two
// (exact) chosen so that |coordinate * boxScale| <= 32768.
// Per child the planes are stored as (lo, hi) PAIRS per axis — x, y, z — so that one 32-bit
// word holds both planes of an axis: the traversal orders them along the ray with one rotate
// (rt_kernels.hip order_planes) instead of a min and a max per axis.
struct alignas(16) Node16 {  // 32 B
  uint16_t box0[6], box1[6];  // box[2 * axis] = lo, box[2 * axis + 1] = hi
  int32_t child[2];           // >= 0: BYTE offset of the child's record (index * 32);
                              // < 0: leaf, ~(byte offset of its first triangle record (first * 48) | count - 1)
};
static_assert(sizeof(Node16) == 32, "packed node record must be 32 bytes");

// The ONE-REQUEST form (round 4): a 16-byte record, so that a node visit costs the CU's vector L1 one divergent
// request instead of two — on trees the caches do not hold, the traversal sits on that unit's request rate
// (DESIGN.md §4.3).  Nodes AND triangle records live in one array of 16-byte slots, cut into blocks of
// 2^blockShift slots (16 KiB by default); slot 0 of every block is its FRAME {origin x, y, z, step} and the 12
// box planes of a record are 8-bit grid coordinates in the frame of the record's own block: plane = origin +
// q * step, step a power of two, lower planes rounded down and upper planes up (the decoded box contains the
// padded float box: the exactness argument of the binary16 form carries over).  Byte order: child 0 x lo, x hi,
// y lo, y hi | child 0 z lo, z hi, child 1 x lo, x hi | child 1 y lo, y hi, z lo, z hi — each word holds two
// (lo, hi) pairs, ordered along the ray by one v_perm_b32 per word.  The two children's ITEMS (an inner child:
// its 16-byte record; a leaf: its 1 or 2 triangle records, 48 bytes each) are adjacent, child 0 first, so ONE
// word locates both: ref = byte offset of child 0's item | bit 0: child 0 is a leaf, bit 1: child 1 is a leaf,
// bit 2 / bit 3: that leaf holds 2 triangles.  Layout: the most-visited top first (greedily by box area from
// the root, as relayoutTop does), then the subtrees depth-first over sibling pairs, so a block is a compact
// piece of the tree (fine grid) and a leaf's triangles sit a few slots from the record that refers to them.
struct alignas(16) Slot16 {
  uint32_t w[4];
};
static_assert(sizeof(Slot16) == 16, "slot must be 16 bytes");
constexpr uint32_t kQ8BlockShift = 10;  // 1,024 slots = 16 KiB per block
constexpr uint32_t kQ8RootOffset = 16;  // the root's record: slot 1 (slot 0 is block 0's frame)

struct alignas(16) TriRec { // 48 B
  float p0[3], e1[3], e2[3];
  uint32_t id;              // global triangle index in reference (mesh, tri) order
  uint32_t mesh;
  uint32_t pad;
};
static_assert(sizeof(TriRec) == 48, "triangle record must be 48 bytes");

constexpr int kMaxDepth = 32;      // traversal stack entries per lane
// LDS budget of the pooled render kernel (rt_kernels.hip plan_persist), which the depth cap is chosen
// against: words per CU available to the waves, words of a wave's ray pool, words per stack row
// (the diagnostic RT_PHASE_TIMING build keeps 256 B of static LDS; the product build has none)
#ifdef RT_PHASE_TIMING
constexpr uint32_t kLdsWordsPerCU = 160u * 1024u / 4u - 64u;
#else
constexpr uint32_t kLdsWordsPerCU = 160u * 1024u / 4u;
#endif
constexpr uint32_t kWavePoolWords = 1128u, kStackRowWords = 64u;
// waves (of at most 16) that fit a CU beside their stacks for a tree of this depth (+1: the sentinel row)
constexpr uint32_t wavesForDepth(int depth) {
  const uint32_t w = kLdsWordsPerCU / ((uint32_t)(depth + 1) * kStackRowWords + kWavePoolWords);
  return w < 16u ? w : 16u;
}
// Round 4: TWENTY waves per CU for the big trees — five workgroups of four waves at 96 VGPRs with the smallest ray pool (168
// words; rt_kernels.hip LT_COMPACT3) — fit while the tree is at most 27 levels deep (the LDS is handed out in units of 1,280
// bytes: 16 x (28 rows x 64 + 168) words = 31,360 B -> 32,000 per workgroup, five of them 160,000 of 163,840).
constexpr uint32_t kWavePoolWordsMin = 168u, kLdsGrainBytes = 1280u;
constexpr bool fitsTwentyWaves(int depth) {
  return 5u * ((((uint32_t)(depth + 1) * kStackRowWords + kWavePoolWordsMin) * 16u + kLdsGrainBytes - 1u) / kLdsGrainBytes * kLdsGrainBytes) <=
         kLdsWordsPerCU * 4u;
}
// Spare levels over the balanced depth.  3 keep SAH within 1 % of the unconstrained tree on small scenes.
// From 19 balanced levels up a level costs waves (256 B of LDS per wave and level), and the trade was
// measured on the lattice scenes.  Round 2, at 14-16 waves: 1 M triangles (19 levels) +2 / +3 / +4 / +5 spare levels = 16 / 15 /
// 15 / 14 waves = 402 / 415 / 416 / 393 ms with 37.5 / 36.7 / 36.7 / 35.0 node visits per ray; 8 M triangles (22 levels) +2 ... +5
// all 14 waves = 73.3 / 71.7 / 71.8 / 63.9 ms, +6 (13 waves) 68.6 ms.  Round 4, with the 20-wave configuration (trees of up to 27
// levels): 1 M triangles at 24 levels (+5) 276.9 ms on 20 waves against 307.7 on 16.  So: as many spare levels, up to 5 and at
// least 3, as keep twenty waves; beyond (more than ~16 M triangles), up to 5 as long as 14 waves (or as many as +2 levels leave)
// still fit.
constexpr int defaultDepthSlack(int levels) {
  if (levels < 19) return 3;
  for (int s = 5; s >= 3; --s)
    if (levels + s < kMaxDepth && fitsTwentyWaves(levels + s)) return s;
  const uint32_t keep = wavesForDepth(levels + 2);
  const uint32_t floorWaves = keep < 14u ? keep : 14u;
  int s = 2;
  while (s < 5 && levels + s + 1 < kMaxDepth && wavesForDepth(levels + s + 1) >= floorWaves) ++s;
  return s;
}
static_assert(defaultDepthSlack(19) == 5 && defaultDepthSlack(22) == 5 && defaultDepthSlack(23) == 4 && fitsTwentyWaves(27) && !fitsTwentyWaves(28), "depth policy");
constexpr uint32_t kTopNodes = 4096;  // nodes [0, kTopNodes) are the most-visited top of the tree (LDS candidates)

inline int32_t encodeLeaf(uint32_t first, uint32_t count) { return ~static_cast<int32_t>((first << 3) | (count - 1)); }

struct Built {
  std::vector<Node> nodes;      // nodes[0] is the root
  std::vector<Node16> nodes16;  // device form of `nodes`
  float boxScale = 1.f;         // power of two applied before the f16 conversion
  std::vector<TriRec> tris;     // leaf order
  std::vector<TriRec> trisRef;  // reference order (brute-force kernel)
  uint32_t maxDepth = 0, leafMax = 2;
  int depthCap = kMaxDepth - 1;  // the cap the tree was built under
  float pad = 0.f;
  float originBound = 0.f;      // ray origins with a larger |coordinate| are outside the padding analysis
  // the one-request form (packQ8): empty unless requested
  std::vector<Slot16> q8;
  uint32_t q8Shift = 0, q8Blocks = 0;
};

// What the device builder (csrc/bvh_gpu.hip) takes from the host before it touches a triangle: validation (throws
// std::runtime_error on an inconsistent description), the leaf size, the HOST builder's depth cap, the box padding, the f16
// plane scale — and per triangle the SIZE key the host builder's splits bin and sweep along beside the three centroid axes
// (-log2 of the longest box edge: the host's libm, so that both builders see the same bits).  One pass shared by `threads`
// threads (0 = as build()).
struct ScenePlan {
  uint32_t leafMax = 2;
  int depthCap = kMaxDepth - 1;  // deepest leaf level the tree may use (root = 0)
  float maxAbs = 0.f, pad = 0.f, originBound = 0.f, boxScale = 1.f;
};
ScenePlan planSceneExact(const rt_scene_desc& scene, uint32_t leafMax, std::vector<float>& sizeKey, uint32_t threads = 0);
uint16_t toHalfDirected(float x, bool up);
float halfToFloat(uint16_t h);

// The TOP of the host builder's tree alone (the hybrid builder, rt_options.bvh_builder = RT_BVH_HYBRID): build() run
// down to ranges of at most `cutoff` triangles — the same split choices (binned SAH with the size axis above 4,096
// triangles, the exact sweep below, the depth budget) — and stopped there.  Every such range ("part") becomes one exact
// subtree on the device (bvh_gpu.hip k_subtree: one workgroup per part), which is where the host build spends most of
// its time.  child refs of `nodes`: >= 0 a top node, < 0 either a leaf (as in Node) or, with bit 30 of ~ref set, the
// part ~ref & 0x3fffffff.
struct TopBuilt {
  struct Part {
    uint32_t b, e;      // range of `order`
    uint32_t depth;     // depth of the part's root node in the whole tree
    uint32_t parent;    // top node that refers to it, and which of its children
    uint32_t slot;
  };
  std::vector<Node> nodes;      // most-visited first (relayoutTop), child 0 = the left range of the split
  std::vector<uint32_t> order;  // triangle ids: the leaf order above the parts, each part's range contiguous
  std::vector<Part> parts;      // ascending by b
  uint32_t leafMax = 2, maxDepth = 0;  // maxDepth: deepest leaf / part root of the top
  int depthCap = kMaxDepth - 1;
  float pad = 0.f, originBound = 0.f, boxScale = 1.f;
};
constexpr uint32_t kPartFlag = 0x40000000u;
void buildTop(const rt_scene_desc& scene, uint32_t leafMax, uint32_t cutoff, TopBuilt& out, uint32_t threads = 0);

// Throws std::runtime_error on an inconsistent scene description.
// threads: builder threads (0 = one per hardware thread, at most 16); the result does not
// depend on it.
void build(const rt_scene_desc& scene, uint32_t leafMax, Built& out, uint32_t threads = 0);
void packNodes(Built& b);  // b.nodes (float boxes) -> b.nodes16 (the device records)
// b.nodes + b.tris -> b.q8 (the unified array of the one-request form).  Throws when the tree cannot be expressed
// (leaves of more than 2 triangles, more than 2 GiB of slots).
void packQ8(Built& b, uint32_t blockShift = kQ8BlockShift);
// One record of b.q8 decoded as the kernels decode it: the two child boxes (float arithmetic of the exact grid
// values) and the child refs in the traversal's form (>= 0: byte offset of the child's record; < 0: leaf,
// ~(byte offset of its first triangle record | count - 1)).
void decodeQ8(const Built& b, uint32_t byteOffset, float lo[2][3], float hi[2][3], int32_t child[2]);
// Final numbering of b.nodes (most-visited top first, see bvh_build.cpp relayoutTop) + packNodes.
void relayoutAndPack(Built& b);
// Measured-cost tuning (bvh_build.cpp): `measure` returns the cost of the tree now in b.nodes (it packs and uploads
// what it needs itself); moves that do not lower it are undone.  Node indices are stable during the search.
struct TuneReport {
  uint32_t probes = 0, accepted = 0;
  double cost0 = 0, cost1 = 0, seconds = 0;
};
TuneReport tuneMeasured(Built& b, const std::function<double()>& measure, double budgetSeconds, uint32_t maxProbes, int verbose);

}  // namespace rtbvh
