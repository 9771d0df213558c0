// rt_lib.hip — kernels + C ABI (include/rtx_hip.h) of the MI355X path-tracing hot path.
// gfx950 only.  Build: raytracing_amd/csrc/Makefile (hipcc --offload-arch=gfx950).
//
// Replaces reference main.rs:64-92 (rayon pixel/sample loop) + main.rs:110-139 (ray_color).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and enums only: the functions are resolved with dlsym when rt_multi_create is called
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "rt_device.h"
#include "scan_layout.h"

using namespace rtd;

namespace {

constexpr int kBlock = 256;          // 4 waves per workgroup share one LDS copy of the scene
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kTile = 8;             // a wave owns an 8x8 pixel tile
// Work units are 8x8 pixels x kUnitSppA samples (3072 items, ~65 scan passes), except for the last kTailSpan (or
// 1/8 of the) samples of a launch, which are cut into kUnitSppB-sample units so that the waves finish together.
constexpr uint32_t kUnitSppA = 48, kUnitSppB = 8, kTailSpan = 128;
constexpr uint32_t kMaxBatch = 1u << 30;  // elements per batch call (grid sizes are computed in 32 bits)
// EXPERIMENT (-DRT_X_TAILORDER; DESIGN.md 7, tools/tail_order_ab.py): measured -1.7 % on configs[2] and -1 % on 500-spp
// renders of configs[1]'s scene, but +-0 on configs[1] itself, +0.5 % on the 4K share and +1.5 % on the two-sphere noise
// scenes, so the product build leaves it out (the code of build_tail_order and its users is dead then: kTailOrderOn folds it away).
#ifdef RT_X_TAILORDER
constexpr bool kTailOrderOn = true;
#else
constexpr bool kTailOrderOn = false;
#endif
// Workspace of rt_render_accum_device_noalloc (the entry point for callers that cannot allocate): the fixed-point
// radiance accumulators of an image of at most RT_NOALLOC_MAX_PIXELS pixels, then the two launch counters.  One copy
// per GPU, held until the process ends (24 MiB + 256 B; + 0.8 MiB in -DRT_X_TAILORDER builds).  In host code this NAME is only a handle for
// hipGetSymbolAddress (rt_create); everything else uses the per-device address kept in rt_ctx::scratch.
constexpr size_t kScratchAccum = (size_t)RT_NOALLOC_MAX_PIXELS * 3;  // u64 elements
// then [0] ray counter, [1] unit counter (+ pad to 32 words), then the tail order of the launch (build_tail_order:
// 16 + 3 * tiles u32; an image of RT_NOALLOC_MAX_PIXELS pixels has at most that many / 16 tiles of 8x8: width >= 2)
constexpr size_t kScratchOrderTiles = kTailOrderOn ? (size_t)RT_NOALLOC_MAX_PIXELS / 16 : 0;  // (experiment builds only)
constexpr size_t kScratchWords = kScratchAccum + 32 + (kTailOrderOn ? (16 + 3 * kScratchOrderTiles) / 2 : 0);
static __device__ __align__(256) unsigned long long g_noalloc_scratch[kScratchWords];

struct RenderArgs {
    DevScene S;
    rt_camera cam;
    uint32_t W, H, tiles_x, n_tiles;
    float inv_wm1, inv_hm1;  // 1 / (W - 1), 1 / (H - 1) in fp32 (main.rs:81-82), divided once on the host
    uint32_t spp_first;      // absolute index of this launch's first sample
    uint32_t spp_count;      // samples per pixel in this launch
    uint32_t spp_a, n_a, spp_b;  // chunk k < n_a covers samples [k*spp_a, (k+1)*spp_a); later chunks are spp_b wide
    uint32_t n_units;        // work units of this launch = n_tiles * chunks; unit u = (tile u % n_tiles, chunk u / n_tiles)
    uint32_t max_depth;
    int32_t antialias;
    uint32_t k0, k1;
    HitConsts hc;            // t_min (main.rs:116), the scan's start and frame origin as f64, ready for the f64 steps
    uint32_t lds_scene_bytes;
    uint32_t acc_bufs;       // tile accumulators per wave in LDS: 2 (current + previous tile) when they fit, else 1
    unsigned long long* accum;  // W*H*3 fixed-point (2^-32) radiance sums
    unsigned long long* rays;
    unsigned int* unit_counter;  // next unclaimed work unit (zeroed before the launch)
    uint32_t* order;             // tile order of the launch's tail (see build_tail_order); NULL: tiles in index order
};

// PERSISTENT waves over a queue of work units.  The grid is exactly the waves the GPU holds at once; a unit is
// one 8x8 pixel tile x one chunk of samples, claimed with one atomic per unit, so all SIMDs stay busy until the
// queue is empty however uneven the path lengths are (no last-round tail of a static grid).  Every lane carries
// TWO paths (the packed scan tests both against each sphere in one instruction stream).  Inside a unit the wave
// runs a private queue of (pixel, sample) items: a path slot whose path ended takes the next item, and when the
// unit's items run out the wave claims the next unit WITHOUT draining: paths of the old unit still in flight
// keep adding into their tile's LDS sums, which the wave holds for one more unit (TWO tile buffers per wave, current
// and previous: a unit lasts ~65 passes, so nearly every straggler ends while its tile is still "previous"); a path
// that outlives even that, or a scene whose noise tables leave LDS room for one buffer only, adds straight into
// the global accumulators (an "orphan").  Radiance is summed in 2^-32 fixed point (order-independent =>
// bit-reproducible images whatever the chunking, tiling or scheduling): LDS per wave and tile, then one global
// atomic per pixel-channel when the tile's buffer is retired.  (With one buffer the in-flight paths of every
// finished unit, ~all 128 slots, were orphans: two thirds of the kernel's HBM traffic.)
// LONG PATHS FIRST in the tail of a launch.  A path of max_depth segments is that many SERIAL passes (5 ms at 100
// bounces in a busy wave); one that starts in the last round of units keeps its wave -- and the launch -- going long after
// the other 3071 waves have left (measured: the mean wave exits 0.3 ms after the queue empties, the last one 3.7 ms).
// Which pixels breed such paths is stable (glass: total internal reflection), so the launch learns it: a path that
// reaches kLongDepth segments is counted on its TILE, and when the queue reaches the small units of the tail
// (RenderArgs::n_a on) the tiles are dealt by class of that count, highest first (kOrderClasses classes: 256 and
// more, 64.., 16.., 8.., 4.., 2.., 1, none): all tail units of a class, chunk-major as before, then the next class.
// The tiles that breed long paths all the time then start theirs while most of the tail's work is still to come, and
// the last rounds of units hold tiles that have bred none in the whole launch so far.  Any order of the units gives the
// same image (order-free sums, counter RNG), so this is scheduling only.
// Workspace `order` (u32): [0] ready flag of this launch's order, [1 .. 1 + kOrderClasses] where each class starts in the
// order (the last entry = n_tiles), [16, 16 + n_tiles) the counts (kept over the launches of a render), then n_tiles
// entries: the tiles in the order they are dealt, then n_tiles entries of builder scratch.  The wave that claims unit tail_order_trigger() -- halfway
// through the big units, long before the tail: one wave's two passes over the counts take ~1 ms for 15 000 tiles and ~8 ms
// for a 4K image -- builds the order; a wave that claims a tail unit before it is ready waits for it -- the builder is resident and running, so the
// wait is finite.  A launch without big units (128 spp or fewer) has nothing to learn from and keeps the index order
// (the host passes no workspace).
constexpr uint32_t kLongDepth = 32;
constexpr int kOrderClasses = 8;
constexpr uint32_t kOrderHeader = 16;
__host__ __device__ inline uint32_t tail_order_trigger(uint32_t n_big, uint32_t) { return n_big / 2u; }
__device__ __forceinline__ int order_class(uint32_t count) {  // 0 = dealt first
    return count >= 256u ? 0 : count >= 64u ? 1 : count >= 16u ? 2 : count >= 8u ? 3 : count >= 4u ? 4 : count >= 2u ? 5 : count >= 1u ? 6 : 7;
}
__device__ __attribute__((noinline)) void build_tail_order(uint32_t* order, uint32_t n_tiles, int lane) {
    const uint32_t* __restrict__ counts = order + kOrderHeader;
    uint32_t* __restrict__ perm = order + kOrderHeader + n_tiles;
    uint32_t* __restrict__ snap = order + kOrderHeader + 2 * n_tiles;  // each tile's class as the first pass saw it: the counts go on growing
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (the counts are device-scope atomics; do not read them from a stale line)
    uint32_t start[kOrderClasses + 1];
#pragma unroll
    for (int k = 0; k <= kOrderClasses; ++k) start[k] = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 64) {  // class sizes (plain loads behind the fence: they pipeline)
        const uint32_t t = t0 + (uint32_t)lane;
        const int cls = t < n_tiles ? order_class(counts[t]) : -1;
        if (t < n_tiles) snap[t] = (uint32_t)cls;
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) start[k + 1] += (uint32_t)__popcll(__ballot(cls == k));
    }
#pragma unroll
    for (int k = 0; k < kOrderClasses; ++k) start[k + 1] += start[k];  // start[k] = first entry of class k; start[last] = n_tiles
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t pos[kOrderClasses];
#pragma unroll
    for (int k = 0; k < kOrderClasses; ++k) pos[k] = start[k];
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += 64) {
        const uint32_t t = t0 + (uint32_t)lane;
        const int cls = t < n_tiles ? (int)snap[t] : -1;  // (this lane's own store of the first pass)
#pragma unroll
        for (int k = 0; k < kOrderClasses; ++k) {
            const unsigned long long bk = __ballot(cls == k);
            if (cls == k) perm[pos[k] + (uint32_t)__popcll(bk & lt)] = t;
            pos[k] += (uint32_t)__popcll(bk);
        }
    }
    if (lane <= kOrderClasses) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k <= kOrderClasses; ++k) v = lane == k ? start[k] : v;
        order[1 + lane] = v;
    }
    __atomic_thread_fence(__ATOMIC_RELEASE);  // every lane's entries before lane 0's flag
    if (lane == 0) __hip_atomic_store(order, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
struct PathState {
    Ray ray;
    f3 thr;
    uint32_t acc_idx;  // where its radiance goes: the LDS element of its pixel's R sum (tile buffer + 3 * pixel in tile), or kOrphan
    uint32_t pix, smp, depth;
    bool alive;
};
constexpr uint32_t kTileAcc = kTile * kTile * 3;  // u64 sums of one tile
constexpr uint32_t kOrphan = 0xFFFFFFFFu;         // PathState::acc_idx of a path whose tile buffer was retired: global atomics
// dynamic LDS after the noise tables, per wave: acc_bufs tile buffers, then the previous tile's geometry (4 words)
__host__ __device__ inline size_t wave_lds_bytes(uint32_t acc_bufs) { return (size_t)acc_bufs * kTileAcc * sizeof(unsigned long long) + 16; }
// EXPERIMENT (-DRT_X_REGROUP=K, not in the product build; DESIGN.md 7, tools/regroup_ab.sh): every K passes the four
// waves of a workgroup re-deal their 512 path slots through LDS, sorted by (direction octant, origin quadrant), dead
// slots last, so that the 128 rays of a wave meet fewer bounds (hit.rs:264-283 culls per ray).  LDS after the waves'
// tile buffers: two sets of 64 bin counters, two sets of four `more` flags, then 19 words per slot (14 of path state,
// 5 of an unfinished trace's record).
#ifdef RT_X_REGROUP
constexpr uint32_t kRgSlots = kBlock * 2, kRgWords = 14, kRgRetryWords = 5;
constexpr size_t kRegroupBytes = (128 + 8 + (size_t)(kRgWords + kRgRetryWords) * kRgSlots) * 4;
#else
constexpr size_t kRegroupBytes = 0;
#endif
__host__ __device__ inline size_t block_lds_bytes(size_t scene_bytes, uint32_t acc_bufs, bool regroups) {
    return scene_bytes + (size_t)kWavesPerBlock * wave_lds_bytes(acc_bufs) + (regroups ? kRegroupBytes : 0);
}

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 3  // 168 VGPRs.  Four waves per SIMD (128 VGPRs) spill ~60 dwords per lane and measure +1 %; two
                           // (a build that takes 174 VGPRs, nothing spilled) measure +24 % on configs[1] and +23 % on
                           // Random: the third wave is needed to keep the VALU pipes busy, a fourth is not
#endif
// kScan: what HitList::hit walks besides the f64 list -- 0: the general scan lists, 1: those and the planar list,
// 2: nothing (a scene of a few spheres, all of them on the f64 list: closest_direct)
template <bool kNoise, bool kTwoBufs, int kScan>
__global__ __launch_bounds__(kBlock, RT_WAVES_PER_EU) void render_kernel(const RenderArgs A_kernarg) {
    constexpr bool kPlanar = kScan == 1, kDirect = kScan == 2;
    // (the variants with noise textures AND scan lists have no registers to spare for it -- 20 to 30 VGPRs spilled -- and no
    // BASELINE scene runs them: they keep the index order)
    constexpr bool kTailOrder = kTailOrderOn && (!kNoise || kDirect);
    // Arguments used in ONE place per pass are not kept in SGPRs across the whole loop (the allocator could only do
    // that by spilling them to VGPR lanes: v_readlane/v_writelane are VALU slots).  The camera and image geometry
    // (refill) and the scan-list fields (scan) are re-read from the kernarg segment (constant address space =>
    // one batch of scalar loads) where they are used; `launder` keeps the optimiser from hoisting those loads.
    typedef const RT_AS4 RenderArgs* ArgsPtr;
    const ArgsPtr K0 = (ArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();  // A_kernarg is the only argument: offset 0
    const RenderArgs& A = A_kernarg;
    extern __shared__ __align__(16) unsigned char lds[];
    constexpr uint32_t kBufs = kTwoBufs ? 2 : 1;  // == RenderArgs::acc_bufs (the host picks the variant)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LdsScene L = stage_scene(A.S, lds, tid, kBlock);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds + A.lds_scene_bytes + wave * wave_lds_bytes(kBufs));  // wave-private
    uint32_t* prev_geo = reinterpret_cast<uint32_t*>(acc + kBufs * kTileAcc);  // {tx0, ty0, tw, npix} of the previous tile
    for (uint32_t e = lane; e < kBufs * kTileAcc; e += 64) acc[e] = 0;
    if (lane < 4) prev_geo[lane] = lane == 2 ? 1u : 0u;
#ifdef RT_X_REGROUP
    constexpr bool kRegroup = !kNoise && !kDirect;
    uint32_t* const rg = reinterpret_cast<uint32_t*>(lds + A.lds_scene_bytes + kWavesPerBlock * wave_lds_bytes(kBufs));
    uint32_t* const rg_stage = rg + 136;
    uint32_t rg_pass = 0, rg_round = 0;
    if (kRegroup) {
        if (tid < 136) rg[tid] = 0;
        __syncthreads();
    }
#else
    constexpr bool kRegroup = false;
#endif

    // current unit (wave-uniform); cur_off: the current tile's buffer (the previous tile's is the other one)
    uint32_t tx0 = 0, ty0 = 0, tw = 1, npix = 0, s_lo = 0;
    uint32_t cur_off = 0;
    uint32_t next = 0, total = 0;
    bool more = true;  // the unit queue may still hold work

    PathState ps[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ps[s].ray = {{0, 0, 0}, {0, 0, 1}, 0};
        ps[s].thr = {1, 1, 1};
        ps[s].pix = ps[s].smp = ps[s].depth = 0;
        ps[s].acc_idx = kOrphan;
        ps[s].alive = false;
    }
    unsigned long long n_rays = 0;
    HitRec h[2];  // closest hits of the previous pass (consumed by the scatter at the top of the next one)
    h[0].sphere = h[1].sphere = -1;

    // a tile's buffer: LDS sums -> global accumulators, LDS cleared (the region is private to this wave)
    auto flush_tile = [&](uint32_t off, uint32_t fx0, uint32_t fy0, uint32_t fw, uint32_t fnpix) {
        // the tile's 3*npix sums are row-major in LDS; lanes take consecutive elements so that the atomics of one
        // instruction fall on consecutive addresses of (at most three) image rows
        const uint32_t row = fw * 3, n = fnpix * 3;
        const ArgsPtr K = launder(K0);
        unsigned long long* const g_accum = K->accum;
        const uint32_t img_w = K->W;
        for (uint32_t e = lane; e < n; e += 64) {
            const unsigned long long q = acc[off + e];
            if (q) {
                const uint32_t ly = e / row, rem = e - ly * row;
                atomicAdd(g_accum + ((size_t)(fy0 + ly) * img_w + fx0) * 3 + rem, q);
                acc[off + e] = 0;
            }
        }
    };

    for (;;) {
        // The unit-queue state is wave-uniform, but the compiler keeps these loop-carried values in VGPRs (nine of them,
        // live across the whole pass) unless each is re-declared uniform once per pass: with them in SGPRs no variant that
        // a BASELINE scene runs spills a VGPR (with them in VGPRs: 5 to 33 spilled, scratch round trips in the scan's cold side).
        next = __builtin_amdgcn_readfirstlane(next); total = __builtin_amdgcn_readfirstlane(total);
        tx0 = __builtin_amdgcn_readfirstlane(tx0); ty0 = __builtin_amdgcn_readfirstlane(ty0);
        tw = __builtin_amdgcn_readfirstlane(tw); npix = __builtin_amdgcn_readfirstlane(npix);
        s_lo = __builtin_amdgcn_readfirstlane(s_lo); cur_off = __builtin_amdgcn_readfirstlane(cur_off);
        more = __builtin_amdgcn_readfirstlane((int)more) != 0;
        // ---- unit queue: claim the next unit once this one has no items left and a slot is free
        RT_MARK("claim");
        if (next >= total && more && __ballot(!ps[0].alive || !ps[1].alive) != 0ull) {
            // retire the OLDER buffer (the previous tile's; with one buffer, the current one): its sums go out, paths
            // of it still in flight become orphans; the current tile becomes the previous one and keeps its buffer
            constexpr bool two = kTwoBufs;
            const uint32_t old_off = two ? kTileAcc - cur_off : 0u;
            if (two) flush_tile(old_off, __builtin_amdgcn_readfirstlane(prev_geo[0]), __builtin_amdgcn_readfirstlane(prev_geo[1]),
                                __builtin_amdgcn_readfirstlane(prev_geo[2]), __builtin_amdgcn_readfirstlane(prev_geo[3]));
            else flush_tile(0u, tx0, ty0, tw, npix);
            if (ps[0].acc_idx - old_off < kTileAcc) ps[0].acc_idx = kOrphan;  // (kOrphan - old_off wraps far above kTileAcc)
            if (ps[1].acc_idx - old_off < kTileAcc) ps[1].acc_idx = kOrphan;
            if (lane == 0) { prev_geo[0] = tx0; prev_geo[1] = ty0; prev_geo[2] = tw; prev_geo[3] = npix; }
            cur_off = old_off;  // the buffer just emptied takes the next tile
            const ArgsPtr K = launder(K0);  // the launch plan: read here (once per ~65 passes), not held in SGPRs
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(K->unit_counter, 1u);
            // The header of the tail order (lane 0: its ready flag, lanes 1 .. kOrderClasses + 1: where each class starts) is
            // fetched alongside the counter, not after it: a claim costs one memory round trip, as before.  Device-scope
            // atomic loads: they bypass the non-coherent cache levels; one instruction, one line: a set flag comes with the
            // starts written before it.
            uint32_t* const order = kTailOrder ? K->order : nullptr;
            uint32_t hdr = 0;
            if (order != nullptr && lane <= kOrderClasses + 1) hdr = __hip_atomic_load(order + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u = __builtin_amdgcn_readfirstlane(u);
            next = 0;
            if (u < K->n_units) {
                uint32_t tile = u % K->n_tiles, chunk = u / K->n_tiles;
                const uint32_t n_big = K->n_a * K->n_tiles;
                if (order != nullptr && u == tail_order_trigger(n_big, K->n_tiles)) build_tail_order(order, K->n_tiles, lane);
                if (order != nullptr && u >= n_big) {  // a tail unit: tiles in the launch's learned order (build_tail_order)
                    const uint32_t n_tiles = K->n_tiles;
                    if (__builtin_amdgcn_readfirstlane(hdr) == 0u) {  // not built yet (rare: the build starts half a launch earlier): wait
                        do {
                            __builtin_amdgcn_s_sleep(16);
                            if (lane <= kOrderClasses + 1) hdr = __hip_atomic_load(order + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } while (__builtin_amdgcn_readfirstlane(hdr) == 0u);
                        __atomic_thread_fence(__ATOMIC_ACQUIRE);
                        if (lane <= kOrderClasses + 1) hdr = __hip_atomic_load(order + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const uint32_t w = u - n_big, tail_chunks = K->n_units / n_tiles - K->n_a;
                    const uint32_t n_counted = (uint32_t)__builtin_amdgcn_readlane((int)hdr, kOrderClasses);  // tiles outside the last class
                    if (n_counted != 0u) {  // (no long path anywhere so far: the order is the index order, nothing to look up)
                        uint32_t idx = 0;
#pragma unroll
                        for (int k = 0; k < kOrderClasses; ++k) {  // the class whose units hold w: units [start * tail_chunks, next * tail_chunks)
                            const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)hdr, 1 + k), s1 = (uint32_t)__builtin_amdgcn_readlane((int)hdr, 2 + k);
                            if (w >= s0 * tail_chunks && w < s1 * tail_chunks) {
                                const uint32_t v = w - s0 * tail_chunks, n_k = s1 - s0;
                                idx = s0 + v % n_k;
                                chunk = K->n_a + v / n_k;
                            }
                        }
                        uint32_t tl = 0;
                        if (lane == 0) tl = __hip_atomic_load(order + kOrderHeader + n_tiles + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        tile = min((uint32_t)__builtin_amdgcn_readfirstlane(tl), n_tiles - 1u);  // (an entry is always a tile index; never index the image with anything else)
                    }
                }
                tx0 = (tile % K->tiles_x) * kTile;
                ty0 = (tile / K->tiles_x) * kTile;
                tw = min((uint32_t)kTile, K->W - tx0);
                npix = tw * min((uint32_t)kTile, K->H - ty0);
                const bool big = chunk < K->n_a;
                s_lo = big ? chunk * K->spp_a : K->n_a * K->spp_a + (chunk - K->n_a) * K->spp_b;
                total = npix * (big ? K->spp_a : min(K->spp_b, K->spp_count - s_lo));
            } else {
                more = false;
                npix = 0;
                total = 0;
            }
        }
        // ---- per slot: a dead slot takes the unit's next item; then ONE Philox call serves whichever the slot needs,
        // the camera sample of a fresh item (dim 0) or the scatter of the hit found in the previous pass
        // (dim 2 + depth).  A wave pays for a call whenever any lane needs one, so merging the two call sites
        // halves the Philox work per pass (it is half of everything outside the scan).
        int noise_tex[2] = {-1, -1};  // noise textures whose value this pass still owes the slots' throughputs
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            PathState& P = ps[s];
            // a slot whose nominee the f64 test rejected in the previous pass (h[s].sphere == kRetry) keeps its ray
            // and is traced again below with that nominee's key as the scan floor: no scatter, no new random numbers
            RT_MARK("refill");
            const bool retry = !kDirect && P.alive && h[s].sphere == kRetry;
            const unsigned long long need = __ballot(!P.alive);
            bool fresh = false;
            uint32_t x = 0, y = 0;
            if (!P.alive) {
                const uint32_t item = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
                if (item < total) {
                    const ArgsPtr K = launder(K0);
                    uint32_t smp_i, pl, ly, lx;
                    if (npix == kTile * kTile) {  // a full tile (wave-uniform): shifts instead of two 32-bit divisions
                        smp_i = item >> 6; pl = item & 63;
                        ly = pl >> 3; lx = pl & 7;
                    } else {
                        smp_i = item / npix; pl = item - smp_i * npix;
                        ly = pl / tw; lx = pl - ly * tw;
                    }
                    P.acc_idx = cur_off + pl * 3;
                    x = tx0 + lx; y = ty0 + ly;
                    P.pix = y * K->W + x;
                    P.smp = K->spp_first + s_lo + smp_i;
                    P.depth = 0;
                    fresh = true;
                }
            }
            next = min(next + (uint32_t)__popcll(need), total);
            if (fresh || (P.alive && !retry)) {
                const ArgsPtr K = launder(K0);  // camera + image geometry: read here, not held across the pass
                RT_MARK("philox");
                uint32_t r[4];
                philox4x32_10(K->k0, K->k1, P.pix, P.smp, fresh ? (uint32_t)DIM_PIXEL : DIM_BOUNCE0 + P.depth, 0u, r);
                RT_MARK("camera");
                if (fresh) {
                    const float ri = K->antialias ? u01(r[0]) : 0.0f;  // main.rs:76-80
                    const float rj = K->antialias ? u01(r[1]) : 0.0f;
                    const float ci = ((float)x + ri) * K->inv_wm1;           // main.rs:81
                    const float cj = 1.0f - ((float)y + rj) * K->inv_hm1;    // main.rs:82
                    float ut = 0.0f;
                    if (K->cam.has_shutter) {
                        uint32_t r1[4];
                        philox4x32_10(K->k0, K->k1, P.pix, P.smp, DIM_TIME, 0u, r1);
                        ut = u01(r1[0]);
                    }
                    P.ray = camera_ray(K->cam, ci, cj, u01(r[2]), u01(r[3]), ut);
                    P.thr = {1.0f, 1.0f, 1.0f};
                    P.alive = true;

                } else {  // the path hit h[s] in the previous pass and goes on (main.rs:117-125)
                    RT_MARK("scatter");
                    const f3 albedo = scatter<kNoise>(K->S, h[s], P.ray, u01(r[0]), u01(r[1]), noise_tex[s]);
                    P.thr = P.thr * albedo;  // main.rs:118
                    ++P.depth;
                }
            }
        }
        RT_MARK("noise");
        if (kNoise) {  // the two slots' noise textures in one packed evaluation at the hit points (= the new ray origins)
            float v0 = 1.0f, v1 = 1.0f;
            noise_pair(launder(K0)->S, L.fused, noise_tex[0], noise_tex[1], ps[0].ray.o, ps[1].ray.o, v0, v1);
            ps[0].thr = ps[0].thr * v0;  // colour = white * value (material.rs:369-373, :529-533)
            ps[1].thr = ps[1].thr * v1;
        }
#ifdef RT_X_REGROUP
        if (kRegroup && ++rg_pass == (uint32_t)(RT_X_REGROUP)) {
            // every wave of the workgroup comes here after the same number of passes (a wave with nothing to do keeps
            // looping) and all of them leave the loop together below: the barriers always meet four waves
            rg_pass = 0;
            const uint32_t par = rg_round & 1u;
            ++rg_round;
            uint32_t* const cnt = rg + par * 64;
            uint32_t* const flg = rg + 128 + par * 4;
            const ArgsPtr K = launder(K0);
            const float cx = (float)K->hc.ox, cz = (float)K->hc.oz;
            uint32_t key[2], rank[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const Ray& r = ps[s].ray;
#ifdef RT_X_REGROUP_NOKEY  // the exchange without the sort (one bin for every live path): what the machinery alone costs
                key[s] = !ps[s].alive ? 32u : 0u; (void)r; (void)cx; (void)cz;
                if (true) { rank[s] = atomicAdd(&cnt[key[s]], 1u); continue; }
#endif
                key[s] = !ps[s].alive ? 32u
                         : (r.d.x > 0.0f ? 1u : 0u) | (r.d.y > 0.0f ? 2u : 0u) | (r.d.z > 0.0f ? 4u : 0u) |
                           (r.o.x > cx ? 8u : 0u) | (r.o.z > cz ? 16u : 0u);
                rank[s] = atomicAdd(&cnt[key[s]], 1u);
            }
            if (lane == 0) flg[wave] = more ? 1u : 0u;
            __syncthreads();
            const uint32_t mine = lane < 33 ? cnt[lane] : 0u;
            uint32_t incl = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off, 64);
                if (lane >= off) incl += t;
            }
            const uint32_t excl = incl - mine;
            const uint32_t n_live = __shfl(excl, 32, 64);
            const bool any_more = (flg[0] | flg[1] | flg[2] | flg[3]) != 0u;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                PathState& P = ps[s];
                const uint32_t pos = __shfl(excl, (int)key[s], 64) + rank[s];
                const bool retry = P.alive && h[s].sphere == kRetry;
                uint32_t* const st = rg_stage + pos;
                st[0 * kRgSlots] = __float_as_uint(P.ray.o.x); st[1 * kRgSlots] = __float_as_uint(P.ray.o.y); st[2 * kRgSlots] = __float_as_uint(P.ray.o.z);
                st[3 * kRgSlots] = __float_as_uint(P.ray.d.x); st[4 * kRgSlots] = __float_as_uint(P.ray.d.y); st[5 * kRgSlots] = __float_as_uint(P.ray.d.z);
                st[6 * kRgSlots] = __float_as_uint(P.ray.time);
                st[7 * kRgSlots] = __float_as_uint(P.thr.x); st[8 * kRgSlots] = __float_as_uint(P.thr.y); st[9 * kRgSlots] = __float_as_uint(P.thr.z);
                st[10 * kRgSlots] = P.pix; st[11 * kRgSlots] = P.smp;
                st[12 * kRgSlots] = P.depth | (P.alive ? 0x80000000u : 0u) | (retry ? 0x40000000u : 0u);
                st[13 * kRgSlots] = (P.acc_idx == kOrphan ? 0x3FFu : P.acc_idx) | ((uint32_t)wave << 10);
                if (retry) {
                    st[14 * kRgSlots] = (uint32_t)h[s].front; st[15 * kRgSlots] = __float_as_uint(h[s].t);
                    st[16 * kRgSlots] = __float_as_uint(h[s].p.x); st[17 * kRgSlots] = __float_as_uint(h[s].p.y); st[18 * kRgSlots] = __float_as_uint(h[s].p.z);
                }
            }
            __syncthreads();
            if (tid < 64) cnt[tid] = 0;  // this set is used again two rounds on, after the two barriers of the next round
            if (n_live == 0 && !any_more) break;  // the same values in all four waves: the workgroup leaves together
            if (rg_round > (1u << 24)) break;      // (never reached; also the same in all four waves)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                PathState& P = ps[s];
                const uint32_t* const st = rg_stage + (uint32_t)wave * 128u + (uint32_t)s * 64u + (uint32_t)lane;
                P.ray.o = {__uint_as_float(st[0 * kRgSlots]), __uint_as_float(st[1 * kRgSlots]), __uint_as_float(st[2 * kRgSlots])};
                P.ray.d = {__uint_as_float(st[3 * kRgSlots]), __uint_as_float(st[4 * kRgSlots]), __uint_as_float(st[5 * kRgSlots])};
                P.ray.time = __uint_as_float(st[6 * kRgSlots]);
                P.thr = {__uint_as_float(st[7 * kRgSlots]), __uint_as_float(st[8 * kRgSlots]), __uint_as_float(st[9 * kRgSlots])};
                P.pix = st[10 * kRgSlots]; P.smp = st[11 * kRgSlots];
                const uint32_t a = st[12 * kRgSlots], b = st[13 * kRgSlots];
                P.depth = a & 0x3FFFFFFFu;
                P.alive = (a & 0x80000000u) != 0u;
                const bool retry = (a & 0x40000000u) != 0u;
                P.acc_idx = ((b >> 10) == (uint32_t)wave && (b & 0x3FFu) != 0x3FFu) ? (b & 0x3FFu) : kOrphan;  // another wave's tile buffer: global sums
                h[s].sphere = retry ? kRetry : -1;
                if (retry) {
                    h[s].front = (int32_t)st[14 * kRgSlots]; h[s].t = __uint_as_float(st[15 * kRgSlots]);
                    h[s].p = {__uint_as_float(st[16 * kRgSlots]), __uint_as_float(st[17 * kRgSlots]), __uint_as_float(st[18 * kRgSlots])};
                }
            }
        }
#endif
        RT_MARK("loopctl");
        const bool any_alive = __ballot(ps[0].alive || ps[1].alive) != 0ull;
        if (!kRegroup && !any_alive && !more) break;  // unit queue empty and every path finished: the only exit, reached by every wave
        if (any_alive) {  // (else next >= total here, so the next pass claims a unit or clears `more`)

        // ---- one ray segment per live path: closest hit (both paths in one packed scan)
        if constexpr (kDirect) {
            RT_MARK("begin");
            const ArgsPtr K = launder(K0);
            const HitConsts C = {K->hc.t_min, K->hc.t_start, K->hc.ox, K->hc.oy, K->hc.oz, K->hc.key_eps0, K->hc.t_start_f};
            closest_direct(K->S, ps[0].ray, ps[1].ray, ps[0].alive, ps[1].alive, C, h[0], h[1]);
        } else
        {
            Hit2 x;
            const bool retry_a = ps[0].alive && h[0].sphere == kRetry, retry_b = ps[1].alive && h[1].sphere == kRetry;
            const Floor fl = {retry_a ? retry_key(h[0]) : -1, retry_b ? retry_key(h[1]) : -1,
                              retry_slot(h[0]), retry_slot(h[1])};
#ifdef RT_X_ALWAYSFLOOR
            const bool any_retry = true;
#else
            const bool any_retry = __ballot(retry_a || retry_b) != 0ull;  // wave-uniform: most passes have none
#endif
            RT_MARK("begin");
            {
                const ArgsPtr K = launder(K0);
                const HitConsts C = {K->hc.t_min, K->hc.t_start, K->hc.ox, K->hc.oy, K->hc.oz, K->hc.key_eps0, K->hc.t_start_f};
                closest_begin<kPlanar>(K->S, ps[0].ray, ps[1].ray, ps[0].alive, ps[1].alive, C, any_retry, h[0], h[1], x);
            }
            {
                const ArgsPtr K = launder(K0);
                const bool sparse_pass = __popcll(__ballot(ps[0].alive)) + __popcll(__ballot(ps[1].alive)) <= kSparseRays;
                ScanLists L;
                L.stat = K->S.lists.stat;
                L.stat_nr2 = K->S.lists.stat_nr2;
                L.stat_bound = K->S.lists.stat_bound;
                L.stat_levels = sparse_pass ? K->S.lists.stat_levels_sparse : K->S.lists.stat_levels;
                L.ns_pad = K->S.lists.ns_pad;
                L.mov_a = K->S.lists.mov_a;
                L.mov_b = K->S.lists.mov_b;
                L.mov_dy = K->S.lists.mov_dy;
                L.mov_bound = K->S.lists.mov_bound;
                L.mov_levels = sparse_pass ? K->S.lists.mov_levels_sparse : K->S.lists.mov_levels;
                L.mov_axes = K->S.lists.mov_axes;
                L.nm_pad = K->S.lists.nm_pad;
                if (kPlanar) {  // (the other variant never reads these)
                    L.plan = K->S.lists.plan;
                    L.plan_nr2 = K->S.lists.plan_nr2;
                    L.plan_bound = K->S.lists.plan_bound;
                    L.plan_axis = K->S.lists.plan_axis;
                    L.plan_levels = sparse_pass ? K->S.lists.plan_levels_sparse : K->S.lists.plan_levels;
                }
                L.np_pad = kPlanar ? K->S.lists.np_pad : 0;  // (slot numbering: read by the moving list's walk too)
                RT_MARK("scan");
                scan_with_floor<kPlanar>(L, fl, x);
            }
            RT_MARK("end");
            {
                const ArgsPtr K = launder(K0);
                const HitConsts C = {K->hc.t_min, K->hc.t_start, K->hc.ox, K->hc.oy, K->hc.oz, K->hc.key_eps0, K->hc.t_start_f};
                closest_end(K->S, ps[0].ray, ps[1].ray, ps[0].alive, ps[1].alive, C, x, h[0], h[1]);
            }
#ifdef RT_STATS
            {  // g_dbg[7]: ray segments that end this pass unfinished (kRetry: passed over or carried on)
                const int n_retry = __popcll(__ballot(ps[0].alive && h[0].sphere == kRetry)) + __popcll(__ballot(ps[1].alive && h[1].sphere == kRetry));
                if (lane == 0 && n_retry) atomicAdd(&g_dbg[7], (unsigned long long)n_retry);
            }
#endif
        }
        // ---- paths that end here: the sky (main.rs:133-138) or the depth budget (main.rs:126-130: black, and the
        // scatter whose ray would never be traced is skipped).  A path that stays alive scatters at the top of the
        // next pass.
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            PathState& P = ps[s];
            RT_MARK("terminate");
            const ArgsPtr K = launder(K0);
            if (P.alive && (kDirect || h[s].sphere != kRetry)) {  // (an unfinished trace: same ray segment again next pass)
                ++n_rays;
                if (h[s].sphere < 0) {
                    P.alive = false;
                    const f3 rad = P.thr * sky(P.ray.d);
                    const float c[3] = {rad.x, rad.y, rad.z};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float v = c[k];
                        v = (v > 0.0f) ? fminf(v, 1048576.0f) : 0.0f;  // NaN / negative -> 0
                        // trunc(v * 2^32) from two 32-bit conversions: the integer part, then the (exact) rest
                        const uint32_t whole = (uint32_t)v;
                        const unsigned long long q = ((unsigned long long)whole << 32) | (uint32_t)((v - (float)whole) * kTwoPow32);
                        if (q) {
                            if (P.acc_idx == kOrphan) atomicAdd(K->accum + (size_t)P.pix * 3 + k, q);
                            else atomicAdd(&acc[P.acc_idx + k], q);
                        }
                    }
                } else if (P.depth + 1 == K->max_depth) {
                    P.alive = false;
                } else if (kTailOrder && P.depth + 1 == kLongDepth && K->order != nullptr) {  // a long path: count it on its tile (build_tail_order)
                    const uint32_t py = P.pix / K->W, px = P.pix - py * K->W;
                    atomicAdd(K->order + kOrderHeader + (py / kTile) * K->tiles_x + px / kTile, 1u);
                }
            }
        }
        }  // any_alive
    }
    // the last tile (the final claim made it the previous one); the current buffer is empty unless the loop is ever left with a tile open
    if (kTwoBufs) flush_tile(kTileAcc - cur_off, __builtin_amdgcn_readfirstlane(prev_geo[0]), __builtin_amdgcn_readfirstlane(prev_geo[1]),
                                    __builtin_amdgcn_readfirstlane(prev_geo[2]), __builtin_amdgcn_readfirstlane(prev_geo[3]));
    flush_tile(cur_off, tx0, ty0, tw, npix);

    // ray-segment counter: wave reduction, one atomic per wave
    for (int off = 32; off > 0; off >>= 1) n_rays += __shfl_down(n_rays, off, 64);
    if (lane == 0 && n_rays) atomicAdd(launder(K0)->rays, n_rays);
}

// the variant a scene runs: with or without the noise textures compiled in, two tile buffers per wave or one
typedef void (*RenderKernel)(const RenderArgs);
template <bool kNoise, bool kTwoBufs>
static RenderKernel render_variant2(int scan) {
    return scan == 2 ? render_kernel<kNoise, kTwoBufs, 2> : scan == 1 ? render_kernel<kNoise, kTwoBufs, 1> : render_kernel<kNoise, kTwoBufs, 0>;
}
static RenderKernel render_variant(bool noise, bool two_bufs, int scan) {
    return noise ? (two_bufs ? render_variant2<true, true>(scan) : render_variant2<true, false>(scan))
                 : (two_bufs ? render_variant2<false, true>(scan) : render_variant2<false, false>(scan));
}

__global__ void finalize_kernel(const unsigned long long* __restrict__ accum, float* __restrict__ out, size_t n,
                                double scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)((double)accum[i] * scale);
}

// the copy-and-add form of the multi-device exchange step (rt_multi_render without RCCL): acc += part
__global__ void add_kernel(float* __restrict__ acc, const float* __restrict__ part, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += part[i];
}

__global__ void scale_kernel(const float* __restrict__ in, float* __restrict__ out, size_t n, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i] * scale;
}

// ---- fixed-batch kernels (parity entry points) --------------------------------------------
__device__ inline void store_hit(const DevScene& S, const HitRec& h, rt_hit* out) {
    rt_hit o;
    memset(&o, 0, sizeof(o));
    if (h.sphere >= 0) {
        o.hit = 1;
        o.sphere_id = h.sphere;
        o.material_id = (int32_t)S.mat_id[h.sphere];
        o.front_face = h.front;
        o.t = h.t;
        o.point[0] = h.p.x; o.point[1] = h.p.y; o.point[2] = h.p.z;
        o.normal[0] = h.n.x; o.normal[1] = h.n.y; o.normal[2] = h.n.z;
    }
    *out = o;
}
// thread i tests rays 2i and 2i+1: the same two-rays-per-lane closest_hit2 the render kernel uses
__global__ __launch_bounds__(kBlock) void intersect_batch_kernel(const DevScene S, const rt_ray* __restrict__ rays,
                                                                 uint32_t n, float t_min, rt_hit* __restrict__ out) {
    const uint32_t i0 = 2 * (blockIdx.x * kBlock + threadIdx.x), i1 = i0 + 1;
    const bool do_a = i0 < n, do_b = i1 < n;
    Ray ra = {{0, 0, 0}, {0, 0, 1}, 0}, rb = ra;
    if (do_a) { const rt_ray q = rays[i0]; ra = {{q.origin[0], q.origin[1], q.origin[2]}, {q.dir[0], q.dir[1], q.dir[2]}, q.time}; }
    if (do_b) { const rt_ray q = rays[i1]; rb = {{q.origin[0], q.origin[1], q.origin[2]}, {q.dir[0], q.dir[1], q.dir[2]}, q.time}; }
    HitRec ha, hb;
    closest_hit2(S, ra, rb, do_a, do_b, t_min, ha, hb);
    if (do_a) store_hit(S, ha, out + i0);
    if (do_b) store_hit(S, hb, out + i1);
}

__global__ __launch_bounds__(kBlock) void scatter_batch_kernel(const DevScene S, const rt_ray* __restrict__ rays,
                                                               const rt_hit* __restrict__ hits,
                                                               const float* __restrict__ u4, uint32_t n,
                                                               rt_scatter* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char lds[];
    const LdsScene L = stage_scene(S, lds, threadIdx.x, kBlock);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    rt_scatter o;
    memset(&o, 0, sizeof(o));
    int noise_tex = -1;
    Ray r = {{0, 0, 0}, {0, 0, 1}, 0};
    f3 a = {0, 0, 0};
    bool hit = false;
    if (i < n) {
        const rt_hit hq = hits[i];
        hit = hq.hit != 0;
        if (hit) {
            const rt_ray q = rays[i];
            r = {{q.origin[0], q.origin[1], q.origin[2]}, {q.dir[0], q.dir[1], q.dir[2]}, q.time};
            HitRec h;
            h.sphere = hq.sphere_id;
            h.front = hq.front_face;
            h.t = hq.t;
            h.p = {hq.point[0], hq.point[1], hq.point[2]};
            h.n = {hq.normal[0], hq.normal[1], hq.normal[2]};
            a = scatter(S, h, r, u4[4 * i], u4[4 * i + 1], noise_tex);
        }
    }
    // noise albedos: a wave-level loop, so every lane gets here (no early return above)
    float v = 1.0f, unused = 1.0f;
    noise_pair(S, L.fused, noise_tex, -1, r.o, r.o, v, unused);
    if (noise_tex >= 0) a = {v, v, v};
    if (hit) {
        o.scattered = 1;
        o.albedo[0] = a.x; o.albedo[1] = a.y; o.albedo[2] = a.z;
        o.origin[0] = r.o.x; o.origin[1] = r.o.y; o.origin[2] = r.o.z;
        o.dir[0] = r.d.x; o.dir[1] = r.d.y; o.dir[2] = r.d.z;
        o.time = r.time;
    }
    if (i < n) out[i] = o;
}

__global__ __launch_bounds__(kBlock) void texture_batch_kernel(const DevScene S, int tex_id,
                                                               const float* __restrict__ p3, uint32_t n,
                                                               float* __restrict__ out3) {
    extern __shared__ __align__(16) unsigned char lds[];
    const LdsScene L = stage_scene(S, lds, threadIdx.x, kBlock);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < n;
    f3 p = {0, 0, 0};
    if (live) p = {p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]};
    const f3 c = texture_value(S, L.fused, live ? tex_id : -1, p);
    if (live) { out3[3 * i] = c.x; out3[3 * i + 1] = c.y; out3[3 * i + 2] = c.z; }
}

__global__ void camera_rays_kernel(const rt_camera cam, const float* __restrict__ ij, const float* __restrict__ u4,
                                   uint32_t n, rt_ray* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ray r = camera_ray(cam, ij[2 * i], ij[2 * i + 1], u4[4 * i], u4[4 * i + 1], u4[4 * i + 2]);
    rt_ray o;
    o.origin[0] = r.o.x; o.origin[1] = r.o.y; o.origin[2] = r.o.z;
    o.dir[0] = r.d.x; o.dir[1] = r.d.y; o.dir[2] = r.d.z;
    o.time = r.time;
    out[i] = o;
}

__global__ void rng_batch_kernel(uint32_t k0, uint32_t k1, const uint32_t* __restrict__ c4, uint32_t n,
                                 uint32_t* __restrict__ out4) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r[4];
    philox4x32_10(k0, k1, c4[4 * i], c4[4 * i + 1], c4[4 * i + 2], c4[4 * i + 3], r);
    out4[4 * i] = r[0]; out4[4 * i + 1] = r[1]; out4[4 * i + 2] = r[2]; out4[4 * i + 3] = r[3];
}

// Screen::encode (screen.rs:26-45): gamma 2, x255.99, truncate, 0RGB.
__global__ void encode_kernel(const float* __restrict__ rgb, uint32_t n, uint32_t* __restrict__ out,
                              uint32_t* __restrict__ n_invalid) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float c[3] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (!(c[k] >= 0.0f && c[k] <= 1.0f)) { bad = true; c[k] = (c[k] > 1.0f) ? 1.0f : 0.0f; }
    if (bad) atomicAdd(n_invalid, 1u);
    // f64 for the last step so truncation agrees with the reference's f64 arithmetic
    const uint32_t r = (uint32_t)(255.99 * sqrt((double)c[0]));
    const uint32_t g = (uint32_t)(255.99 * sqrt((double)c[1]));
    const uint32_t b = (uint32_t)(255.99 * sqrt((double)c[2]));
    out[i] = (r << 16) | (g << 8) | b;
}

thread_local std::string g_create_error;

}  // namespace

// ---------------------------------------------------------------------------------- context
struct rt_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    std::string err;
    // scene
    bool has_scene = false;
    DevScene S{};
    std::vector<void*> scene_allocs;
    size_t lds_scene = 0;
    // render workspace
    unsigned long long* accum = nullptr;
    size_t accum_cap = 0;  // elements
    uint32_t* order = nullptr;  // tail order workspace of the allocating entry points (build_tail_order)
    size_t order_cap = 0;       // tiles
    uint32_t order_tiles = 0;   // tiles of the last render that used `order` (rt_debug_tail_order_stats)
    bool dbg_no_tail_order = false;
    unsigned long long* rays = nullptr;
    float* out = nullptr;
    size_t out_cap = 0;
    int cu_count = 256;
    bool has_noise = false;  // the uploaded scene references a value/Perlin noise texture
    int blocks_per_cu = 0;   // resident workgroups of the scene's render-kernel variant (computed at upload)
    uint32_t acc_bufs = 1;   // LDS tile buffers per wave (RenderArgs::acc_bufs; computed at upload)
    // this device's copy of g_noalloc_scratch (hipGetSymbolAddress at rt_create) and a pinned word for the ray count
    unsigned long long* scratch = nullptr;
    unsigned long long* pinned_rays = nullptr;
    // rt_debug_* (include/rtx_hip_debug.h): work-unit sizes forced by a test, scan lists kept in world order
    uint32_t dbg_unit_a = 0, dbg_unit_b = 0;
    int32_t dbg_tail_span = -1;
    bool dbg_world_order = false;
    bool dbg_no_planar = false;
    int scan_mode = 0;        // render_kernel's kScan for the uploaded scene: 0 general lists, 1 + planar list, 2 no lists
    bool dbg_no_direct = false;
    int dbg_tile_bufs = 0;
};

namespace {

int fail(rt_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}
#define RT_HIP(ctx, call)                                                                            \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(ctx, RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)

template <class T>
int dev_copy(rt_ctx* c, const T* host, size_t n, const T** out, bool track = true) {
    void* d = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    RT_HIP(c, hipMalloc(&d, bytes));
    if (track) c->scene_allocs.push_back(d);
    if (n) RT_HIP(c, hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(d);
    return RT_OK;
}

void free_scene(rt_ctx* c) {
    for (void* p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->has_scene = false;
}

struct TmpBuf {  // device scratch for the batch entry points
    void* p = nullptr;
    ~TmpBuf() { if (p) (void)hipFree(p); }
};

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// FusedNoise image of one noise table for one kind (rt_device.h).  The gradient of hash value h is the TABLE of
// material.rs:431-450, case h & 15 (cases 9, 11 and 15 are all -y-z and 14 is -y+z there: reproduced, not "fixed").
void build_fused(const rt_noise_table& tb, bool perlin, FusedNoise& out) {
    static const float kGrad[16][3] = {
        {1, 1, 0}, {-1, 1, 0}, {1, -1, 0}, {-1, -1, 0},   // 0-3:   +-x +-y
        {1, 0, 1}, {-1, 0, 1}, {1, 0, -1}, {-1, 0, -1},   // 4-7:   +-x +-z
        {0, 1, 1}, {0, -1, -1}, {0, 1, -1}, {0, -1, -1},  // 8-11:  y+z, -y-z, y-z, -y-z
        {1, 1, 0}, {-1, 1, 0}, {0, -1, 1}, {0, -1, -1}};  // 12-15: x+y, -x+y, -y+z, -y-z
    for (int i = 0; i < 512; ++i) {
        out.perms[i] = tb.perms[i];
        const int h = tb.perms[i];
        if (perlin) out.g[i] = make_float4(kGrad[h & 15][0], kGrad[h & 15][1], kGrad[h & 15][2], 0.0f);
        else out.g[i] = make_float4(0.0f, 0.0f, 0.0f, tb.randoms[h]);
    }
}

int check_ready(rt_ctx* c) {
    if (!c) return RT_ERR_BAD_ARG;
    RT_HIP(c, hipSetDevice(c->device));
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded (call rt_upload_scene first)");
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_create(int device_ordinal, rt_ctx** out) {
    if (!out) return fail(nullptr, RT_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device_ordinal < 0 || device_ordinal >= n)
        return fail(nullptr, RT_ERR_BAD_ARG, "device ordinal out of range");
    rt_ctx* c = new rt_ctx();
    c->device = device_ordinal;
    hipError_t e = hipSetDevice(device_ordinal);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreate(&c->ev2);
    if (e == hipSuccess) e = hipEventCreate(&c->ev3);
    if (e == hipSuccess) e = hipMalloc((void**)&c->rays, 2 * sizeof(unsigned long long));  // [0] ray counter, [1] unit counter
    if (e == hipSuccess) e = hipHostMalloc((void**)&c->pinned_rays, 2 * sizeof(unsigned long long), hipHostMallocDefault);
    // this device's address of the no-alloc workspace (each GPU holds its own copy of the array)
    if (e == hipSuccess) e = hipGetSymbolAddress((void**)&c->scratch, HIP_SYMBOL(g_noalloc_scratch));
    if (e == hipSuccess && !c->scratch) e = hipErrorInvalidSymbol;
    if (e != hipSuccess) {
        g_create_error = std::string("rt_create: ") + hipGetErrorString(e);
        rt_destroy(c);
        return RT_ERR_HIP;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) {
        c->cu_count = prop.multiProcessorCount;
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            g_create_error = std::string("rt_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
            rt_destroy(c);
            return RT_ERR_NO_DEVICE;
        }
    }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_scene(c);
    if (c->accum) (void)hipFree(c->accum);
    if (c->order) (void)hipFree(c->order);
    if (c->rays) (void)hipFree(c->rays);
    if (c->out) (void)hipFree(c->out);
    if (c->pinned_rays) (void)hipHostFree(c->pinned_rays);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev2) (void)hipEventDestroy(c->ev2);
    if (c->ev3) (void)hipEventDestroy(c->ev3);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rt_upload_scene(rt_ctx* c, const rt_scene* s) {
    if (!c || !s) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    RT_HIP(c, hipSetDevice(c->device));
    // n_spheres == 0 is a valid world (an empty HitList, hit.rs:100-110: every ray sees the sky)
    if (s->n_spheres > RT_MAX_SPHERES) return fail(c, RT_ERR_SCENE_TOO_LARGE, "more than RT_MAX_SPHERES spheres");
    if ((s->n_spheres && s->n_materials == 0) || s->n_materials > RT_MAX_MATERIALS) return fail(c, RT_ERR_BAD_ARG, "bad n_materials");
    if (s->n_textures > RT_MAX_TEXTURES) return fail(c, RT_ERR_SCENE_TOO_LARGE, "more than RT_MAX_TEXTURES textures");
    if (s->n_noise_tables > RT_MAX_NOISE_TABLES) return fail(c, RT_ERR_SCENE_TOO_LARGE, "more than RT_MAX_NOISE_TABLES noise tables");
    if (s->n_spheres && (!s->c0x || !s->c0y || !s->c0z || !s->dcx || !s->dcy || !s->dcz || !s->radius || !s->material_id))
        return fail(c, RT_ERR_BAD_ARG, "NULL scene array");
    if (s->n_materials && !s->materials) return fail(c, RT_ERR_BAD_ARG, "NULL scene array");
    if ((s->n_textures && !s->textures) || (s->n_noise_tables && !s->noise_tables))
        return fail(c, RT_ERR_BAD_ARG, "NULL texture / noise table array");
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const rt_material& m = s->materials[i];
        if (m.type < RT_MAT_LAMBERTIAN || m.type > RT_MAT_DIELECTRIC) return fail(c, RT_ERR_BAD_ARG, "unknown material type");
        if (m.type == RT_MAT_LAMBERTIAN && (m.texture_id < 0 || (uint32_t)m.texture_id >= s->n_textures))
            return fail(c, RT_ERR_BAD_ARG, "Lambertian texture_id out of range");
        if (m.type == RT_MAT_DIELECTRIC && !(m.ior > 0.0f)) return fail(c, RT_ERR_BAD_ARG, "Dielectric ior must be > 0");
    }
    for (uint32_t t = 0; t < s->n_noise_tables; ++t)
        for (int i = 0; i < 512; ++i)
            if (s->noise_tables[t].perms[i] > 255) return fail(c, RT_ERR_BAD_ARG, "noise perms entry > 255");
    std::vector<DevTexture> texs(s->n_textures);
    std::vector<FusedNoise> fused;                  // one image per (noise table, kind) pair in use
    std::vector<std::pair<int32_t, int32_t>> fused_key;
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const rt_texture& t = s->textures[i];
        if (t.kind < RT_TEX_SOLID || t.kind > RT_TEX_PERLIN_NOISE) return fail(c, RT_ERR_BAD_ARG, "unknown texture kind");
        texs[i].t = t;
        texs[i].inv_fractal_max = 1.0f;
        texs[i].fused_id = -1;
        texs[i].pad = 0;
        texs[i].freq_turns = (float)((double)t.freq / 6.283185307179586476925);
        if (t.kind >= RT_TEX_VALUE_NOISE) {
            if (t.table_id < 0 || (uint32_t)t.table_id >= s->n_noise_tables) return fail(c, RT_ERR_BAD_ARG, "noise table_id out of range");
            if (t.op < RT_NOISE_PLAIN || t.op > RT_NOISE_TURBULENCE) return fail(c, RT_ERR_BAD_ARG, "unknown noise op");
            if (t.op != RT_NOISE_PLAIN) {
                // the asserts of material.rs:185-186 / :212-213
                if (t.layers <= 0) return fail(c, RT_ERR_BAD_ARG, "fBm: Can't have 0 layers.");
                if (!(0.0f < t.gain && t.gain < 1.0f)) return fail(c, RT_ERR_BAD_ARG, "fBm: Gain must be in range (0, 1).");
                double g = t.gain, pw = 1.0;
                for (int l = 0; l < t.layers; ++l) pw *= g;
                texs[i].inv_fractal_max = (float)((1.0 - g) / (1.0 - pw));
            }
            const std::pair<int32_t, int32_t> key(t.table_id, t.kind);
            size_t f = std::find(fused_key.begin(), fused_key.end(), key) - fused_key.begin();
            if (f == fused_key.size()) {
                if (fused.size() == (size_t)kMaxFused)
                    return fail(c, RT_ERR_SCENE_TOO_LARGE, "more than 6 (noise table, noise kind) pairs in use: their LDS images do not fit");
                fused_key.push_back(key);
                fused.emplace_back();
                build_fused(s->noise_tables[t.table_id], t.kind == RT_TEX_PERLIN_NOISE, fused.back());
            }
            texs[i].fused_id = (int32_t)f;
        }
    }

    // the layout of the world for the scan: device-free code (scan_layout.h), tested on the CPU
    ScanLayout lay;
    {
        LayoutOptions opt;
        opt.world_order = c->dbg_world_order; opt.no_planar = c->dbg_no_planar; opt.no_direct = c->dbg_no_direct;
        std::string lerr;
        const int lrc = lay.build(s, opt, lerr);
        if (lrc != RT_OK) return fail(c, lrc, lerr);
    }
    auto as4 = [](const std::vector<F4>& v) { return reinterpret_cast<const float4*>(v.data()); };
    const std::vector<F4>&stat = lay.stat, &plan = lay.plan, &mov_a = lay.mov_a, &mov_b = lay.mov_b;
    const std::vector<int32_t>&scan_id = lay.scan_id, &big_id = lay.big_id;
    const std::vector<float>& inv_rad = lay.inv_rad;
    const size_t lds_need = lds_scene_bytes((int)fused.size());
    if (lds_need > 64 * 1024) return fail(c, RT_ERR_SCENE_TOO_LARGE, "scene does not fit the 64 KiB LDS budget");

    free_scene(c);
    DevScene D{};
    int rc;
#define UP(expr) if ((rc = (expr)) != RT_OK) { free_scene(c); return rc; }
    UP(dev_copy(c, as4(stat), stat.size(), &D.lists.stat));
    UP(dev_copy(c, lay.stat_nr2.data(), lay.stat_nr2.size(), &D.lists.stat_nr2));
    UP(dev_copy(c, as4(lay.stat_bound), lay.stat_bound.size(), &D.lists.stat_bound));
    UP(dev_copy(c, as4(lay.mov_bound), lay.mov_bound.size(), &D.lists.mov_bound));
    UP(dev_copy(c, as4(plan), plan.size(), &D.lists.plan));
    UP(dev_copy(c, lay.plan_nr2.data(), lay.plan_nr2.size(), &D.lists.plan_nr2));
    UP(dev_copy(c, as4(lay.plan_bound), lay.plan_bound.size(), &D.lists.plan_bound));
    UP(dev_copy(c, as4(mov_a), mov_a.size(), &D.lists.mov_a));
    UP(dev_copy(c, as4(mov_b), mov_b.size(), &D.lists.mov_b));
    {
        std::vector<float> mov_dy(mov_b.size());
        for (size_t k = 0; k < mov_b.size(); ++k) mov_dy[k] = mov_b[k].y;
        UP(dev_copy(c, mov_dy.data(), mov_dy.size(), &D.lists.mov_dy));
    }
    UP(dev_copy(c, scan_id.data(), scan_id.size(), &D.scan_id));
    UP(dev_copy(c, big_id.data(), big_id.size(), &D.big_id));
    std::vector<BigSphere> big(big_id.size());
    for (size_t k = 0; k < big_id.size(); ++k) {
        const int32_t i = big_id[k];
        big[k] = {(double)s->c0x[i], (double)s->c0y[i], (double)s->c0z[i], (double)s->radius[i] * (double)s->radius[i],
                  (double)s->dcx[i], (double)s->dcy[i], (double)s->dcz[i], (double)inv_rad[i]};
    }
    UP(dev_copy(c, big.data(), big.size(), &D.big));
    {  // the 16-byte records of the f64 decision (scan-slot order) and of the hit record (world order)
        std::vector<float4> slot_a(scan_id.size(), make_float4(0, 0, 0, 0)), slot_dc(slot_a), rec_a(s->n_spheres), rec_dc(s->n_spheres);
        for (size_t k = 0; k < scan_id.size(); ++k) {
            const int32_t i = scan_id[k];
            if (i < 0) continue;  // a pad: never nominated
            slot_a[k] = make_float4(s->c0x[i], s->c0y[i], s->c0z[i], s->radius[i]);
            slot_dc[k] = make_float4(s->dcx[i], s->dcy[i], s->dcz[i], 0.0f);
        }
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            rec_a[i] = make_float4(s->c0x[i], s->c0y[i], s->c0z[i], inv_rad[i]);
            rec_dc[i] = make_float4(s->dcx[i], s->dcy[i], s->dcz[i], 0.0f);
        }
        UP(dev_copy(c, slot_a.data(), slot_a.size(), &D.slot_a));
        UP(dev_copy(c, slot_dc.data(), slot_dc.size(), &D.slot_dc));
        UP(dev_copy(c, rec_a.data(), rec_a.size(), &D.rec_a));
        UP(dev_copy(c, rec_dc.data(), rec_dc.size(), &D.rec_dc));
    }
    UP(dev_copy(c, s->material_id, (size_t)s->n_spheres, &D.mat_id));
    std::vector<ShadeRec> shade(s->n_spheres);
    for (uint32_t i = 0; i < s->n_spheres; ++i) {
        const rt_material& m = s->materials[s->material_id[i]];
        ShadeRec& q = shade[i];
        q = ShadeRec{m.type, m.fuzz, m.ior, -1, {1.0f, 1.0f, 1.0f}, RT_TEX_SOLID, {1.0f, 1.0f, 1.0f}, 0.0f};
        if (m.type == RT_MAT_METAL) {
            for (int k = 0; k < 3; ++k) q.a[k] = m.albedo[k];
        } else if (m.type == RT_MAT_LAMBERTIAN) {
            const rt_texture& t = s->textures[m.texture_id];
            if (t.kind == RT_TEX_SOLID || t.kind == RT_TEX_CHECKER) {
                for (int k = 0; k < 3; ++k) { q.a[k] = t.color_a[k]; q.b[k] = t.color_b[k]; }
                q.kind = t.kind;
                q.freq_turns = texs[m.texture_id].freq_turns;
            } else {
                q.noise_tex = m.texture_id;
            }
        }
    }
    UP(dev_copy(c, shade.data(), shade.size(), &D.shade));
    UP(dev_copy(c, texs.data(), texs.size(), &D.texs));
    UP(dev_copy(c, fused.data(), fused.size(), &D.fused));
#undef UP
    D.lists.ox = lay.org[0]; D.lists.oy = lay.org[1]; D.lists.oz = lay.org[2];
    D.lists.cmax = lay.cmax;
    D.lists.key_eps0 = lay.key_eps0;
    D.lists.stat_levels = lay.stat_levels;
    D.lists.mov_levels = lay.mov_levels;
    D.lists.plan_levels = lay.plan_levels;
    D.lists.stat_levels_sparse = lay.stat_levels_sparse;
    D.lists.mov_levels_sparse = lay.mov_levels_sparse;
    D.lists.plan_levels_sparse = lay.plan_levels_sparse;
    D.lists.mov_axes = lay.mov_axes;
    D.lists.plan_axis = lay.plan_axis;
    D.lists.plan_val = lay.plan_val;
    D.lists.ns_pad = (int32_t)stat.size();
    D.lists.np_pad = (int32_t)plan.size();
    D.lists.nm_pad = (int32_t)mov_a.size();
    c->scan_mode = (stat.empty() && plan.empty() && mov_a.empty()) ? 2 : !plan.empty() ? 1 : 0;
    D.n_big = (int32_t)big_id.size();
    D.n_fused = (int32_t)fused.size();
    D.n_spheres = (int32_t)s->n_spheres;
    D.n_textures = (int32_t)s->n_textures;
    D.any_moving = lay.any_moving;
    c->S = D;
    c->lds_scene = align16(lds_need);
    c->has_noise = false;
    for (uint32_t i = 0; i < s->n_textures; ++i) c->has_noise |= s->textures[i].kind >= RT_TEX_VALUE_NOISE;
    // resident workgroups per CU of the variant this scene runs (the persistent grid is exactly that many), and
    // the tile buffers per wave: two, unless the second one would not fit the 64 KiB of dynamic LDS next to the
    // noise tables or would cost a resident workgroup
    int occ[3] = {0, 0, 0};
    for (uint32_t bufs = 1; bufs <= 2; ++bufs) {
        const size_t lds_bytes = block_lds_bytes(c->lds_scene, bufs, !c->has_noise && c->scan_mode != 2);
        if (lds_bytes > 65536) continue;
        RT_HIP(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ[bufs], render_variant(c->has_noise, bufs == 2, c->scan_mode), kBlock, lds_bytes));
    }
    c->acc_bufs = (occ[2] >= occ[1] && occ[2] > 0) ? 2 : 1;
    if (c->dbg_tile_bufs && occ[c->dbg_tile_bufs] > 0) c->acc_bufs = (uint32_t)c->dbg_tile_bufs;  // rt_debug_set_tile_buffers
    c->blocks_per_cu = occ[c->acc_bufs];
#ifdef RT_X_REGROUP
    fprintf(stderr, "RT_X_REGROUP=%d: %d workgroups per CU, %zu bytes of LDS each\n", (int)(RT_X_REGROUP), c->blocks_per_cu,
            block_lds_bytes(c->lds_scene, c->acc_bufs, !c->has_noise && c->scan_mode != 2));
#endif
    c->has_scene = true;
    return RT_OK;
}

// Argument checks shared by the render entry points; nothing is allocated or touched before they pass.
static int check_render_args(rt_ctx* c, const rt_camera* cam, const rt_params* p, const void* out) {
    int rc = check_ready(c);
    if (rc != RT_OK) return rc;
    if (!cam || !p || !out) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    if (p->width < 2 || p->height < 2) return fail(c, RT_ERR_BAD_ARG, "width and height must be >= 2 (i = x/(W-1), main.rs:81)");
    if (p->width > 65536 || p->height > 65536 || (uint64_t)p->width * p->height > (1ull << 28))
        return fail(c, RT_ERR_BAD_ARG, "image larger than 65536 on a side or 2^28 pixels");
    if (p->spp == 0 || p->max_depth == 0) return fail(c, RT_ERR_BAD_ARG, "spp and max_depth are NonZero in the reference (config.rs:52,56)");
    if ((uint64_t)p->spp_offset + p->spp > 0xFFFFFFFFull) return fail(c, RT_ERR_BAD_ARG, "spp_offset + spp overflows 32 bits");
    if (!(p->t_min >= 0.0f)) return fail(c, RT_ERR_BAD_ARG, "t_min must be >= 0");
    if (p->reserved != 0) return fail(c, RT_ERR_BAD_ARG, "rt_params.reserved must be 0");
    return RT_OK;
}

// The render proper: clears the workspace, launches the path-trace kernel(s) and the finalize on the ctx stream.
// `accum` (W*H*3 u64) and `counters` ([0] rays, [1] unit queue) are the workspace: the ctx's own allocation or
// this device's static scratch array.  Allocates nothing.  sync = false leaves the work in flight (multi-device
// callers launch on every device first); render_finish waits and fills the stats.
static int render_launch(rt_ctx* c, const rt_camera* cam, const rt_params* p, unsigned long long* accum,
                         unsigned long long* counters, uint32_t* order, float* d_out, double out_scale, uint32_t* n_launches) {
    const size_t n_px = (size_t)p->width * p->height, n_el = n_px * 3;
    if (!kTailOrderOn || c->dbg_no_tail_order) order = nullptr;  // (an experiment: -DRT_X_TAILORDER) rt_debug_set_tail_order
    RT_HIP(c, hipEventRecord(c->ev0, c->stream));
    RT_HIP(c, hipMemsetAsync(accum, 0, n_el * sizeof(unsigned long long), c->stream));
    if (order == c->order) c->order_tiles = order ? ((p->width + kTile - 1) / kTile) * ((p->height + kTile - 1) / kTile) : 0;
    if (order)  // ready flag, class starts and the tiles' counts (kept over the launches of this render)
        RT_HIP(c, hipMemsetAsync(order, 0, (kOrderHeader + (size_t)((p->width + kTile - 1) / kTile) * ((p->height + kTile - 1) / kTile)) * sizeof(uint32_t), c->stream));
    RT_HIP(c, hipMemsetAsync(counters, 0, sizeof(unsigned long long), c->stream));

    RenderArgs A{};
    A.S = c->S;
    A.cam = *cam;
    A.W = p->width; A.H = p->height;
    A.inv_wm1 = 1.0f / ((float)p->width - 1.0f);  // IEEE fp32 division, as the kernel did per sample
    A.inv_hm1 = 1.0f / ((float)p->height - 1.0f);
    A.tiles_x = (p->width + kTile - 1) / kTile;
    A.n_tiles = A.tiles_x * ((p->height + kTile - 1) / kTile);
    A.max_depth = p->max_depth;
    A.antialias = p->antialias;
    A.k0 = (uint32_t)p->seed; A.k1 = (uint32_t)(p->seed >> 32);
    A.hc = make_hit_consts(c->S.lists, p->t_min);
    A.lds_scene_bytes = (uint32_t)c->lds_scene;
    A.acc_bufs = c->acc_bufs;
    A.accum = accum;
    A.rays = counters;
    const size_t lds_bytes = block_lds_bytes(c->lds_scene, c->acc_bufs, !c->has_noise && c->scan_mode != 2);

    // Launch plan.  A launch renders `spp_l` samples per pixel as n_tiles * n_levels work units that a grid of
    // persistent waves (as many as the GPU holds at once) pulls from a counter.  Launches are capped at 2^31
    // samples so no single kernel runs for minutes.
    const uint64_t max_samples_per_launch = 1ull << 31;
    uint32_t spp_per_launch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(p->spp, max_samples_per_launch / n_px));
    const bool noise = c->has_noise;  // noise-free scenes run the variant with the noise textures compiled out
    if (c->blocks_per_cu < 1) return fail(c, RT_ERR_HIP, "render kernel does not fit a compute unit");
    const uint32_t resident_blocks = (uint32_t)c->cu_count * (uint32_t)c->blocks_per_cu;
    uint32_t done = 0, launches = 0;
    RT_HIP(c, hipEventRecord(c->ev1, c->stream));
    while (done < p->spp) {
        const uint32_t spp_l = std::min(spp_per_launch, p->spp - done);
        A.spp_first = p->spp_offset + done;
        A.spp_count = spp_l;
        // Unit sizes (see kUnitSppA): chunk-major unit order means the small units are the last ones claimed.
        // Measured on configs[1] (tools/unit_sweep.sh): uniform 16 / 24 / 32-spp units 262.6 / 264.6 / 266.5 ms,
        // 125 spp +9 %, 500 spp +30 % (tiles differ 3x in cost, so the end-of-launch spread is several units);
        // 48-spp units with a 128-spp tail of 8-spp units 261.8 ms, 40/8/96 260.6 ms, 64/8/64 269 ms; a schedule of
        // halving sizes cut the accumulator traffic 6x but cost 4-6 %.
        uint32_t ua = kUnitSppA, ub = kUnitSppB, span = std::max(kTailSpan, spp_l / 8);
        if (c->dbg_unit_a) ua = c->dbg_unit_a;  // rt_debug_set_work_units (tests, sweeps): the image does not depend on these
        if (c->dbg_unit_b) ub = c->dbg_unit_b;
        if (c->dbg_tail_span >= 0) span = (uint32_t)c->dbg_tail_span;
        span = std::min(span, spp_l);
        A.spp_a = ua; A.spp_b = ub;
        A.n_a = (spp_l - span) / ua;
        const uint32_t rest = spp_l - A.n_a * ua;
        const uint32_t chunks = A.n_a + (rest + ub - 1) / ub;
        const uint64_t n_units = (uint64_t)A.n_tiles * chunks;
        if (n_units > 0xFFFFFFF0ull) return fail(c, RT_ERR_BAD_ARG, "too many work units in one launch");
        A.n_units = (uint32_t)n_units;
        A.unit_counter = reinterpret_cast<unsigned int*>(counters + 1);
        RT_HIP(c, hipMemsetAsync(counters + 1, 0, sizeof(unsigned long long), c->stream));
        A.order = A.n_a > 0 ? order : nullptr;  // (no big units: nothing to learn the order from)
        if (order && launches > 0) RT_HIP(c, hipMemsetAsync(order, 0, sizeof(uint32_t), c->stream));  // this launch builds its own order
        const uint32_t blocks = (uint32_t)std::min<uint64_t>(resident_blocks, (n_units + kWavesPerBlock - 1) / kWavesPerBlock);
        hipLaunchKernelGGL(render_variant(noise, c->acc_bufs == 2, c->scan_mode), dim3(blocks), dim3(kBlock), lds_bytes, c->stream, A);
        RT_HIP(c, hipGetLastError());
        done += spp_l;
        ++launches;
    }
    RT_HIP(c, hipEventRecord(c->ev2, c->stream));
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, c->stream, accum, d_out, n_el, out_scale);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipMemcpyAsync(c->pinned_rays, counters, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipEventRecord(c->ev3, c->stream));
    *n_launches = launches;
    return RT_OK;
}

static int render_finish(rt_ctx* c, const rt_params* p, uint32_t launches, rt_stats* stats) {
    RT_HIP(c, hipSetDevice(c->device));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->rays_traced = c->pinned_rays[0];
        stats->samples = (uint64_t)p->width * p->height * p->spp;
        RT_HIP(c, hipEventElapsedTime(&stats->kernel_ms, c->ev1, c->ev2));
        RT_HIP(c, hipEventElapsedTime(&stats->total_ms, c->ev0, c->ev3));
        stats->n_launches = launches;
    }
    return RT_OK;
}

// ctx-owned accumulators, grown on demand (the allocating entry points)
static int ensure_accum(rt_ctx* c, size_t n_el) {
    if (c->accum_cap >= n_el) return RT_OK;
    if (c->accum) (void)hipFree(c->accum);
    c->accum = nullptr; c->accum_cap = 0;
    if (hipMalloc((void**)&c->accum, n_el * sizeof(unsigned long long)) != hipSuccess)
        return fail(c, RT_ERR_ALLOC, "cannot allocate the radiance accumulators");
    c->accum_cap = n_el;
    return RT_OK;
}
static uint32_t tiles_of(const rt_params* p) { return ((p->width + kTile - 1) / kTile) * ((p->height + kTile - 1) / kTile); }
static int ensure_order(rt_ctx* c, size_t n_tiles) {
    if (!kTailOrderOn || c->order_cap >= n_tiles) return RT_OK;
    if (c->order) (void)hipFree(c->order);
    c->order = nullptr; c->order_cap = 0;
    if (hipMalloc((void**)&c->order, (kOrderHeader + 3 * n_tiles) * sizeof(uint32_t)) != hipSuccess)
        return fail(c, RT_ERR_ALLOC, "cannot allocate the tile order of the launch tail");
    c->order_cap = n_tiles;
    return RT_OK;
}

static const double kFixedToFloat = 1.0 / 4294967296.0;

int rt_render_accum_device(rt_ctx* c, const rt_camera* cam, const rt_params* p, float* d_sum_rgb, rt_stats* stats) {
    int rc = check_render_args(c, cam, p, d_sum_rgb);
    if (rc != RT_OK) return rc;
    if ((rc = ensure_accum(c, (size_t)p->width * p->height * 3)) != RT_OK) return rc;
    if ((rc = ensure_order(c, tiles_of(p))) != RT_OK) return rc;
    uint32_t launches = 0;
    if ((rc = render_launch(c, cam, p, c->accum, c->rays, c->order, d_sum_rgb, kFixedToFloat, &launches)) != RT_OK) return rc;
    return render_finish(c, p, launches, stats);
}

// Same render with NO allocation: the accumulators and counters live in this device's static scratch array.
int rt_render_accum_device_noalloc(rt_ctx* c, const rt_camera* cam, const rt_params* p, float* d_sum_rgb,
                                   rt_stats* stats) {
    int rc = check_render_args(c, cam, p, d_sum_rgb);
    if (rc != RT_OK) return rc;
    if ((uint64_t)p->width * p->height > RT_NOALLOC_MAX_PIXELS)
        return fail(c, RT_ERR_BAD_ARG, "rt_render_accum_device_noalloc: image larger than RT_NOALLOC_MAX_PIXELS pixels "
                                       "(the static workspace is fixed; use rt_render_accum_device)");
    uint32_t launches = 0;
    // every call on this GPU shares the array and it keeps the last call's contents: render_launch clears the
    // W*H*3 accumulators it uses and both counters before the kernel reads them
    uint32_t* const order = tiles_of(p) <= kScratchOrderTiles ? reinterpret_cast<uint32_t*>(c->scratch + kScratchAccum + 32) : nullptr;
    if ((rc = render_launch(c, cam, p, c->scratch, c->scratch + kScratchAccum, order, d_sum_rgb, kFixedToFloat, &launches)) != RT_OK)
        return rc;
    return render_finish(c, p, launches, stats);
}

int rt_render(rt_ctx* c, const rt_camera* cam, const rt_params* p, float* out_rgb, rt_stats* stats) {
    int rc = check_render_args(c, cam, p, out_rgb);  // before any allocation: a rejected call leaves the ctx as it was
    if (rc != RT_OK) return rc;
    const size_t n_el = (size_t)p->width * p->height * 3;
    if ((rc = ensure_accum(c, n_el)) != RT_OK) return rc;
    if ((rc = ensure_order(c, tiles_of(p))) != RT_OK) return rc;
    if (c->out_cap < n_el) {
        if (c->out) (void)hipFree(c->out);
        c->out = nullptr; c->out_cap = 0;
        if (hipMalloc((void**)&c->out, n_el * sizeof(float)) != hipSuccess)
            return fail(c, RT_ERR_ALLOC, "cannot allocate the output image");
        c->out_cap = n_el;
    }
    uint32_t launches = 0;
    if ((rc = render_launch(c, cam, p, c->accum, c->rays, c->order, c->out, kFixedToFloat / (double)p->spp, &launches)) != RT_OK) return rc;
    if ((rc = render_finish(c, p, launches, stats)) != RT_OK) return rc;
    RT_HIP(c, hipMemcpy(out_rgb, c->out, n_el * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---- debug hooks (include/rtx_hip_debug.h): not part of the drop-in ABI ----------------------------
int rt_debug_set_work_units(rt_ctx* c, uint32_t unit_a, uint32_t unit_b, int32_t tail_span) {
    if (!c) return RT_ERR_BAD_ARG;
    c->dbg_unit_a = unit_a; c->dbg_unit_b = unit_b; c->dbg_tail_span = tail_span;
    return RT_OK;
}
int rt_debug_set_planar_list(rt_ctx* c, int on) {
    if (!c) return RT_ERR_BAD_ARG;
    c->dbg_no_planar = on == 0;
    return RT_OK;
}
int rt_debug_scan_layout(rt_ctx* c, int32_t out[8]) {
    if (!c || !out) return RT_ERR_BAD_ARG;
    if (!c->has_scene) return fail(c, RT_ERR_NO_SCENE, "no scene uploaded");
    const ScanLists& L = c->S.lists;
    const int32_t v[8] = {c->S.n_big, L.ns_pad, L.np_pad, L.nm_pad, L.np_pad > 0 ? L.plan_axis : -1,
                          L.nm_pad > 0 ? L.mov_axes : 0, (int32_t)c->acc_bufs, c->has_noise ? 1 : 0};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return RT_OK;
}
int rt_debug_set_tail_order(rt_ctx* c, int on) {
    if (!c) return RT_ERR_BAD_ARG;
    c->dbg_no_tail_order = on == 0;
    return RT_OK;
}
int rt_debug_tail_order_stats(rt_ctx* c, uint32_t out[3]) {
    if (!c || !out) return RT_ERR_BAD_ARG;
    out[0] = out[1] = out[2] = 0;
    if (!c->order || c->order_tiles == 0) return RT_OK;
    RT_HIP(c, hipSetDevice(c->device));
    const uint32_t n = c->order_tiles;
    std::vector<uint32_t> h(kOrderHeader + 2 * (size_t)n);
    RT_HIP(c, hipMemcpy(h.data(), c->order, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    out[0] = h[0];
    out[1] = h[kOrderClasses];  // where the last class (no long path seen) starts = the tiles that have bred one
    if (h[0]) {  // the order must be a permutation of the tiles, and the class starts ascending up to n
        std::vector<char> seen(n, 0);
        bool ok = h[1] == 0 && h[1 + kOrderClasses] == n;
        for (int k = 0; k < kOrderClasses; ++k) ok = ok && h[1 + k] <= h[2 + k];
        for (uint32_t i = 0; i < n && ok; ++i) {
            const uint32_t t = h[kOrderHeader + n + i];
            ok = t < n && !seen[t];
            if (ok) seen[t] = 1;
        }
        out[2] = ok ? 1u : 0u;
    }
    return RT_OK;
}
int rt_debug_set_direct(rt_ctx* c, int on) {
    if (!c) return RT_ERR_BAD_ARG;
    c->dbg_no_direct = on == 0;
    return RT_OK;
}
int rt_debug_set_tile_buffers(rt_ctx* c, int n) {
    if (!c || n < 0 || n > 2) return RT_ERR_BAD_ARG;
    c->dbg_tile_bufs = n;
    return RT_OK;
}
int rt_debug_set_scan_order(rt_ctx* c, int world_order) {
    if (!c) return RT_ERR_BAD_ARG;
    c->dbg_world_order = world_order != 0;
    return RT_OK;
}

// ---- batch entry points ---------------------------------------------------------------------
#define RT_TMP(ctx, buf, bytes) RT_HIP(ctx, hipMalloc(&(buf).p, (bytes) ? (bytes) : 1))

int rt_intersect_batch(rt_ctx* c, const rt_ray* rays, uint32_t n, float t_min, rt_hit* out) {
    int rc = check_ready(c);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    if (n > kMaxBatch) return fail(c, RT_ERR_BAD_ARG, "batch larger than 2^30 elements");
    if (!rays || !out) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    TmpBuf dr, dh;
    RT_TMP(c, dr, n * sizeof(rt_ray));
    RT_TMP(c, dh, n * sizeof(rt_hit));
    RT_HIP(c, hipMemcpy(dr.p, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(intersect_batch_kernel, dim3((n + 2 * kBlock - 1) / (2 * kBlock)), dim3(kBlock), 0, c->stream,
                       c->S, (const rt_ray*)dr.p, n, t_min, (rt_hit*)dh.p);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out, dh.p, n * sizeof(rt_hit), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_scatter_batch(rt_ctx* c, const rt_ray* rays, const rt_hit* hits, const float* u4, uint32_t n, rt_scatter* out) {
    int rc = check_ready(c);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    if (n > kMaxBatch) return fail(c, RT_ERR_BAD_ARG, "batch larger than 2^30 elements");
    if (!rays || !hits || !u4 || !out) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    for (uint32_t i = 0; i < n; ++i)
        if (hits[i].hit && (hits[i].sphere_id < 0 || hits[i].sphere_id >= c->S.n_spheres))
            return fail(c, RT_ERR_BAD_ARG, "hit.sphere_id out of range");
    TmpBuf dr, dh, du, dout;
    RT_TMP(c, dr, n * sizeof(rt_ray));
    RT_TMP(c, dh, n * sizeof(rt_hit));
    RT_TMP(c, du, n * 4 * sizeof(float));
    RT_TMP(c, dout, n * sizeof(rt_scatter));
    RT_HIP(c, hipMemcpy(dr.p, rays, n * sizeof(rt_ray), hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(dh.p, hits, n * sizeof(rt_hit), hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(du.p, u4, n * 4 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(scatter_batch_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), c->lds_scene, c->stream,
                       c->S, (const rt_ray*)dr.p, (const rt_hit*)dh.p, (const float*)du.p, n, (rt_scatter*)dout.p);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out, dout.p, n * sizeof(rt_scatter), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_texture_batch(rt_ctx* c, int32_t tex_id, const float* p3, uint32_t n, float* out3) {
    int rc = check_ready(c);
    if (rc != RT_OK) return rc;
    if (n == 0) return RT_OK;
    if (n > kMaxBatch) return fail(c, RT_ERR_BAD_ARG, "batch larger than 2^30 elements");
    if (!p3 || !out3) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    TmpBuf dp, dout;
    RT_TMP(c, dp, n * 3 * sizeof(float));
    RT_TMP(c, dout, n * 3 * sizeof(float));
    if (tex_id < 0 || tex_id >= c->S.n_textures) return fail(c, RT_ERR_BAD_ARG, "texture_id out of range");
    RT_HIP(c, hipMemcpy(dp.p, p3, n * 3 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(texture_batch_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), c->lds_scene, c->stream,
                       c->S, tex_id, (const float*)dp.p, n, (float*)dout.p);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out3, dout.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_camera_rays(rt_ctx* c, const rt_camera* cam, const float* ij2, const float* u4, uint32_t n, rt_ray* out) {
    if (!c) return RT_ERR_BAD_ARG;
    RT_HIP(c, hipSetDevice(c->device));
    if (n == 0) return RT_OK;
    if (n > kMaxBatch) return fail(c, RT_ERR_BAD_ARG, "batch larger than 2^30 elements");
    if (!cam || !ij2 || !u4 || !out) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    TmpBuf dij, du, dout;
    RT_TMP(c, dij, n * 2 * sizeof(float));
    RT_TMP(c, du, n * 4 * sizeof(float));
    RT_TMP(c, dout, n * sizeof(rt_ray));
    RT_HIP(c, hipMemcpy(dij.p, ij2, n * 2 * sizeof(float), hipMemcpyHostToDevice));
    RT_HIP(c, hipMemcpy(du.p, u4, n * 4 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(camera_rays_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, *cam, (const float*)dij.p,
                       (const float*)du.p, n, (rt_ray*)dout.p);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out, dout.p, n * sizeof(rt_ray), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_rng_batch(rt_ctx* c, uint64_t seed, const uint32_t* c4, uint32_t n, uint32_t* out4) {
    if (!c) return RT_ERR_BAD_ARG;
    RT_HIP(c, hipSetDevice(c->device));
    if (n == 0) return RT_OK;
    if (n > kMaxBatch) return fail(c, RT_ERR_BAD_ARG, "batch larger than 2^30 elements");
    if (!c4 || !out4) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    TmpBuf dc, dout;
    RT_TMP(c, dc, n * 4 * sizeof(uint32_t));
    RT_TMP(c, dout, n * 4 * sizeof(uint32_t));
    RT_HIP(c, hipMemcpy(dc.p, c4, n * 4 * sizeof(uint32_t), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rng_batch_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, (uint32_t)seed,
                       (uint32_t)(seed >> 32), (const uint32_t*)dc.p, n, (uint32_t*)dout.p);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out4, dout.p, n * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_encode(rt_ctx* c, const float* rgb, uint32_t n_pixels, uint32_t* out_0rgb, uint32_t* n_invalid) {
    if (!c) return RT_ERR_BAD_ARG;
    RT_HIP(c, hipSetDevice(c->device));
    if (n_invalid) *n_invalid = 0;
    if (n_pixels == 0) return RT_OK;
    if (n_pixels > kMaxBatch) return fail(c, RT_ERR_BAD_ARG, "more than 2^30 pixels");
    if (!rgb || !out_0rgb) return fail(c, RT_ERR_BAD_ARG, "NULL argument");
    TmpBuf din, dout, dcnt;
    RT_TMP(c, din, (size_t)n_pixels * 3 * sizeof(float));
    RT_TMP(c, dout, (size_t)n_pixels * sizeof(uint32_t));
    RT_TMP(c, dcnt, sizeof(uint32_t));
    RT_HIP(c, hipMemcpy(din.p, rgb, (size_t)n_pixels * 3 * sizeof(float), hipMemcpyHostToDevice));
    RT_HIP(c, hipMemsetAsync(dcnt.p, 0, sizeof(uint32_t), c->stream));
    hipLaunchKernelGGL(encode_kernel, dim3((n_pixels + 255) / 256), dim3(256), 0, c->stream, (const float*)din.p,
                       n_pixels, (uint32_t*)dout.p, (uint32_t*)dcnt.p);
    RT_HIP(c, hipGetLastError());
    RT_HIP(c, hipStreamSynchronize(c->stream));
    RT_HIP(c, hipMemcpy(out_0rgb, dout.p, (size_t)n_pixels * sizeof(uint32_t), hipMemcpyDeviceToHost));
    uint32_t cnt = 0;
    RT_HIP(c, hipMemcpy(&cnt, dcnt.p, sizeof(cnt), hipMemcpyDeviceToHost));
    if (n_invalid) *n_invalid = cnt;
    return RT_OK;
}

}  // extern "C"


// ---------------------------------------------------------------------------------- multi-device
// One process, n devices: samples-per-pixel are partitioned over the devices (device g renders the sample indices
// rt_partition_spp gives it, for every pixel: disjoint counter-RNG streams, scene replicated), then the fp32
// radiance-sum buffers are summed onto the first device with ONE ncclReduce over xGMI.  RCCL is bound at run time
// (dlopen) so that single-device callers do not need it installed.
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
#define RT_SYM(field, sym) \
    if (!(field = reinterpret_cast<decltype(field)>(dlsym(lib, sym)))) { err = std::string("librccl lacks ") + sym; return false; }
        RT_SYM(CommInitAll, "ncclCommInitAll") RT_SYM(CommDestroy, "ncclCommDestroy") RT_SYM(Reduce, "ncclReduce")
        RT_SYM(GroupStart, "ncclGroupStart") RT_SYM(GroupEnd, "ncclGroupEnd") RT_SYM(GetErrorString, "ncclGetErrorString")
#undef RT_SYM
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_multi_create_error;
}  // namespace

struct rt_multi {
    std::vector<rt_ctx*> ctx;      // one per device, in the caller's order; ctx[0] is the reduce root
    std::vector<ncclComm_t> comm;  // empty when the exchange step runs without RCCL (use_rccl == false)
    std::vector<float*> dsum;      // per device: W*H*3 radiance sums of its share
    float* stage = nullptr;        // copy-and-add exchange: a part's sums on the root device, W*H*3 floats
    size_t dsum_cap = 0;
    bool use_rccl = true;
    std::string err;
};
namespace {
// rt_debug_multi (include/rtx_hip_debug.h): bit 0 = a device ordinal may be listed more than once, bit 1 = never use
// RCCL.  Process-wide, read by rt_multi_create.
int g_multi_debug = 0;
}  // namespace

extern "C" {

int rt_partition_spp(uint32_t total_spp, uint32_t n_parts, uint32_t part, uint32_t* first, uint32_t* count) {
    if (!first || !count || n_parts == 0 || part >= n_parts) return RT_ERR_BAD_ARG;
    const uint32_t base = total_spp / n_parts, rem = total_spp % n_parts;
    *count = base + (part < rem ? 1u : 0u);          // the remainder goes to the lowest parts
    *first = part * base + std::min(part, rem);
    return RT_OK;
}

const char* rt_multi_last_error(const rt_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

void rt_multi_destroy(rt_multi* m) {
    if (!m) return;
    for (size_t i = 0; i < m->ctx.size(); ++i) {
        if (m->ctx[i]) (void)hipSetDevice(m->ctx[i]->device);
        if (i < m->comm.size() && m->comm[i] && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(m->comm[i]);
        if (i < m->dsum.size() && m->dsum[i]) (void)hipFree(m->dsum[i]);
    }
    if (m->stage && !m->ctx.empty() && m->ctx[0]) { (void)hipSetDevice(m->ctx[0]->device); (void)hipFree(m->stage); }
    for (rt_ctx* c : m->ctx) rt_destroy(c);
    delete m;
}

int rt_multi_create(const int* device_ordinals, int n_devices, rt_multi** out) {
    auto bad = [&](int code, const std::string& msg) { g_multi_create_error = msg; return code; };
    if (!out) return bad(RT_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    if (!device_ordinals || n_devices < 1 || n_devices > RT_MAX_DEVICES) return bad(RT_ERR_BAD_ARG, "need 1..RT_MAX_DEVICES device ordinals");
    bool repeated = false;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j) repeated |= device_ordinals[i] == device_ordinals[j];
    if (repeated && !(g_multi_debug & 1)) return bad(RT_ERR_BAD_ARG, "a device ordinal is listed twice");
    rt_multi* m = new rt_multi();
    m->ctx.assign(n_devices, nullptr);
    m->dsum.assign(n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
        const int rc = rt_create(device_ordinals[i], &m->ctx[i]);
        if (rc != RT_OK) {
            g_multi_create_error = std::string("device ") + std::to_string(device_ordinals[i]) + ": " + rt_last_error(nullptr);
            rt_multi_destroy(m);
            return rc;
        }
    }
    // The exchange step is one ncclReduce over xGMI.  Where a communicator cannot be formed -- librccl
    // missing, ncclCommInitAll failing, or (debug hook) one device listed several times, which RCCL refuses -- the
    // parts are copied to the root device and added there (hipMemcpyPeerAsync + add_kernel): same sums, n - 1 copies.
    m->use_rccl = !repeated && !(g_multi_debug & 2);
    if (m->use_rccl) {
        std::string err;
        if (!g_rccl.load(err)) {
            m->use_rccl = false;
        } else {
            m->comm.assign(n_devices, nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, device_ordinals);  // rank i = device_ordinals[i]
            if (r != ncclSuccess) { m->comm.clear(); m->use_rccl = false; }
        }
    }
    *out = m;
    return RT_OK;
}

int rt_multi_upload_scene(rt_multi* m, const rt_scene* scene) {
    if (!m) return RT_ERR_BAD_ARG;
    for (rt_ctx* c : m->ctx) {  // the scene (tens of KiB) is replicated
        const int rc = rt_upload_scene(c, scene);
        if (rc != RT_OK) { m->err = c ? c->err : "NULL ctx"; return rc; }
    }
    return RT_OK;
}

int rt_multi_render(rt_multi* m, const rt_camera* cam, const rt_params* p, float* out_rgb, rt_multi_stats* stats) {
    if (!m) return RT_ERR_BAD_ARG;
    const int n = (int)m->ctx.size();
    // An error return must not leave work in flight on the devices already launched: wait for every stream first.
    auto drain = [&]() {
        for (int i = 0; i < n; ++i) {
            (void)hipSetDevice(m->ctx[i]->device);
            (void)hipStreamSynchronize(m->ctx[i]->stream);
        }
    };
    auto fail_from = [&](rt_ctx* c, int rc) { m->err = c->err; drain(); return rc; };
    auto fail_msg = [&](int rc, const std::string& msg) { m->err = msg; drain(); return rc; };
    int rc = check_render_args(m->ctx[0], cam, p, out_rgb);
    if (rc != RT_OK) { m->err = m->ctx[0]->err; return rc; }
    const size_t n_el = (size_t)p->width * p->height * 3;
    rt_ctx* root = m->ctx[0];
    if (m->dsum_cap < n_el) {
        for (int i = 0; i < n; ++i) {
            (void)hipSetDevice(m->ctx[i]->device);
            if (m->dsum[i]) (void)hipFree(m->dsum[i]);
            m->dsum[i] = nullptr;
        }
        (void)hipSetDevice(root->device);
        if (m->stage) (void)hipFree(m->stage);
        m->stage = nullptr;
        m->dsum_cap = 0;
        for (int i = 0; i < n; ++i) {
            (void)hipSetDevice(m->ctx[i]->device);
            if (hipMalloc((void**)&m->dsum[i], n_el * sizeof(float)) != hipSuccess) {
                m->err = "cannot allocate the per-device radiance sums";
                return RT_ERR_ALLOC;
            }
        }
        if (!m->use_rccl && n > 1) {
            (void)hipSetDevice(root->device);
            if (hipMalloc((void**)&m->stage, n_el * sizeof(float)) != hipSuccess) {
                m->err = "cannot allocate the staging buffer of the copy-and-add exchange";
                return RT_ERR_ALLOC;
            }
        }
        m->dsum_cap = n_el;
    }
    // 1. every device renders its share; all launches are queued before any is waited for
    std::vector<rt_params> share(n, *p);
    std::vector<uint32_t> launches(n, 0);
    for (int i = 0; i < n; ++i) {
        rt_ctx* c = m->ctx[i];
        uint32_t first = 0, count = 0;
        (void)rt_partition_spp(p->spp, (uint32_t)n, (uint32_t)i, &first, &count);
        share[i].spp = count;
        share[i].spp_offset = p->spp_offset + first;
        if ((rc = check_ready(c)) != RT_OK) return fail_from(c, rc);
        if (count == 0) {  // more devices than samples: this one contributes zeros to the sum
            if (hipMemsetAsync(m->dsum[i], 0, n_el * sizeof(float), c->stream) != hipSuccess) return fail_msg(RT_ERR_HIP, "hipMemsetAsync failed");
            continue;
        }
        // (several entries of one device -- the debug hook -- share that device's GPU but not their workspaces)
        if ((rc = ensure_accum(c, n_el)) != RT_OK) return fail_from(c, rc);
        if ((rc = ensure_order(c, tiles_of(&share[i]))) != RT_OK) return fail_from(c, rc);
        if ((rc = render_launch(c, cam, &share[i], c->accum, c->rays, c->order, m->dsum[i], kFixedToFloat, &launches[i])) != RT_OK)
            return fail_from(c, rc);
    }
    rt_multi_stats st;
    std::memset(&st, 0, sizeof(st));
    st.n_devices = (uint32_t)n;
    st.exchange = m->use_rccl ? 0u : 1u;
    st.samples = (uint64_t)p->width * p->height * p->spp;
    for (int i = 0; i < n; ++i) {
        rt_ctx* c = m->ctx[i];
        if (share[i].spp == 0) {
            (void)hipSetDevice(c->device);
            if (hipStreamSynchronize(c->stream) != hipSuccess) return fail_msg(RT_ERR_HIP, "hipStreamSynchronize failed");
            continue;
        }
        rt_stats s1;
        if ((rc = render_finish(c, &share[i], launches[i], &s1)) != RT_OK) return fail_from(c, rc);
        st.rays_traced += s1.rays_traced;
        st.kernel_ms = std::max(st.kernel_ms, s1.kernel_ms);
        st.device_kernel_ms[i] = s1.kernel_ms;
    }
    // 2. the exchange step: the W*H*3 fp32 sums of every part onto ctx[0] (every render has finished: step 1 waited)
    (void)hipSetDevice(root->device);
    if (hipEventRecord(root->ev0, root->stream) != hipSuccess) return fail_msg(RT_ERR_HIP, "hipEventRecord failed");
    if (m->use_rccl) {  // ONE sum-reduce over xGMI, in place on the root
        ncclResult_t r = g_rccl.GroupStart();
        for (int i = 0; i < n && r == ncclSuccess; ++i)
            r = g_rccl.Reduce(m->dsum[i], m->dsum[i], n_el, ncclFloat, ncclSum, 0, m->comm[i], m->ctx[i]->stream);
        const ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return fail_msg(RT_ERR_RCCL, std::string("ncclReduce: ") + g_rccl.GetErrorString(r));
    } else {  // no communicator: copy each part to the root device and add it there, in device order
        for (int i = 1; i < n; ++i) {
            if (share[i].spp == 0) continue;  // zeros
            if (hipMemcpyPeerAsync(m->stage, root->device, m->dsum[i], m->ctx[i]->device, n_el * sizeof(float), root->stream) != hipSuccess)
                return fail_msg(RT_ERR_HIP, "hipMemcpyPeerAsync failed");
            hipLaunchKernelGGL(add_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, root->stream, m->dsum[0], (const float*)m->stage, n_el);
            if (hipGetLastError() != hipSuccess) return fail_msg(RT_ERR_HIP, "add_kernel launch failed");
        }
    }
    (void)hipSetDevice(root->device);
    if (hipEventRecord(root->ev1, root->stream) != hipSuccess) return fail_msg(RT_ERR_HIP, "hipEventRecord failed");
    // 3. mean over all samples on the root, back to the host
    if (root->out_cap < n_el) {
        if (root->out) (void)hipFree(root->out);
        root->out = nullptr; root->out_cap = 0;
        if (hipMalloc((void**)&root->out, n_el * sizeof(float)) != hipSuccess) return fail_msg(RT_ERR_ALLOC, "cannot allocate the output image");
        root->out_cap = n_el;
    }
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, root->stream, (const float*)m->dsum[0],
                       root->out, n_el, (float)(1.0 / (double)p->spp));
    for (int i = 0; i < n; ++i) {
        (void)hipSetDevice(m->ctx[i]->device);
        if (hipStreamSynchronize(m->ctx[i]->stream) != hipSuccess) return fail_msg(RT_ERR_HIP, "hipStreamSynchronize failed after the exchange step");
    }
    (void)hipSetDevice(root->device);
    if (hipMemcpy(out_rgb, root->out, n_el * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return fail_msg(RT_ERR_HIP, "hipMemcpy failed");
    (void)hipEventElapsedTime(&st.reduce_ms, root->ev0, root->ev1);
    if (stats) *stats = st;
    return RT_OK;
}

int rt_debug_multi(int flags) {
    if (flags < 0 || flags > 3) return RT_ERR_BAD_ARG;
    g_multi_debug = flags;
    return RT_OK;
}

}  // extern "C"

#ifdef RT_STATS
extern "C" int rt_debug_counters(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rtd::g_dbg), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rtd::g_dbg), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
