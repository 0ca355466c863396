// rt_render.hip -- the gfx950 render kernels: sampler + integrators of the reference as ONE
// persistent launch per image.
//
// Reference control flow (CPU): RandomSampler::sample_image runs spp serial passes, each a rayon
// par_chunks_mut over 10 000-pixel chunks with a barrier, each pixel calling
// Mis/NaiveIntegrator::get_colour; the caller folds every pass into a running mean
// (samplers/random_sampler.rs:10-99, integrators/mod.rs:22-78, integrators/mis.rs:7-157,
// src/main.rs:175-191).
//
// MI355X design: there are no passes.  A lane OWNS a pixel for all of its samples: it draws the
// camera ray, walks the path, folds the sample into the pixel's running mean in registers
// (`mean += (sample - mean) / i`, the reference's own update, in sample order -- so the mean is
// bit-identical to the reference's accumulation order), and writes the pixel once.
//
// Paths last 1..49 bounces, a ray visits 3..500 BVH nodes, about half of all shading events end
// the sample: 64 lanes in lockstep would mostly wait for each other.  So the kernel is a per-lane
// state machine whose steps ("phases") are small, and every loop iteration the WAVE votes
// (__ballot + popcount) and executes the one phase most of its lanes are waiting for; the other
// lanes keep their state in registers / their traversal stack in LDS and are served by a later
// iteration:
//   GEN      new sample: seed the stream, camera ray, start a closest-hit walk
//   NODE     one inner-node step of a BVH walk (fetch 64-B node, two slab tests, push/pop)
//   LEAF     intersect the primitives of one leaf
//   SHADE    closest hit known: emission, MIS weight, Russian roulette, sample finished?   (mis.rs:17-33,50-86)
//   LIGHT    sample_lights: pick sky / light primitive, build the shadow ray, start an any-hit walk (mis.rs:95-157)
//   SCATTER  light contribution, BSDF sample -> next ray, start a closest-hit walk          (mis.rs:39-49)
// For tiny trees (a walk is 1-3 steps) the vote is coarser: TRACE = GEN + walk + SHADE and
// LIGHT = LIGHT + shadow walk + SCATTER are voted as two super-phases (constexpr FINE below).
// NODE and LEAF are shared by path rays and shadow rays (a per-lane flag selects closest / any
// hit), so bounce rays of some lanes and shadow rays of others traverse together.  A lane whose
// pixel is finished pulls the next one from a global queue with one wave-aggregated atomic.
// Paths never read each other's state and each owns its random stream (seed, pixel, sample), so
// the schedule cannot change a pixel: results are bit-identical to the serial reference order.
//
// Scene data is read-only and shared: BVH / primitives through L1/L2 from HBM, the sky's CDF rows
// staged once per workgroup into LDS next to the per-lane traversal stacks.  HBM write traffic is
// 12 bytes per pixel for the whole render.  No MFMA: nothing here is a dense contraction.
#include "rt_shade.h"

namespace rt {

enum : int {
	PH_GEN = 0,
	PH_NODE = 1,
	PH_LEAF = 2,
	PH_SHADE = 3,
	PH_LIGHT = 4,
	PH_SCATTER = 5,
	PH_NARROW = 6,     // fine schedule only: an irregular ray (rt_intersect.h) runs the two-child walk to its end
	PH_COUNT = 7,
	PH_NEED_PIXEL = 7, // served at the top of every iteration (one atomic per wave), not voted
	PH_DONE = 8,
	PH_WALK = 9        // fine schedule with the exchange only: a walk waiting to start (between two iterations, never voted)
};

// coarse schedule: the LIGHT super-phase runs once this many lanes of the wave wait for it
// (rtweekend1 MIS, same box: 32 -> 152 ms, 40 -> 138 ms, 48 / 52 / 56 -> 135-137 ms)
#ifndef RT_PQ_SPLIT
#define RT_PQ_SPLIT 1
#endif
#ifndef RT_LIGHT_PHASE_THRESHOLD
#define RT_LIGHT_PHASE_THRESHOLD 48
#endif
constexpr uint32_t kLightPhaseThreshold = RT_LIGHT_PHASE_THRESHOLD;
// ... or, short of that many, once the paths that wait for it have waited this many lane-iterations between them.  The lane
// threshold alone strands them in a wave whose other lanes only ever regenerate: a wave that holds a few cheap (sky) items next to
// expensive ones never collects 48 waiting lanes, and the waiting ones sit idle until the cheap items run out.  Whole-pixel items
// hid this (a wave then takes a tile at a time); with sample_split a wave whose lanes finish at different times pulls a block of
// cheap items and hands them to a few lanes at a time (stats build, config 2 at S = 16: 37 % of PRIMARY iterations ran with
// 17 - 24 lanes while 40 - 47 waited).  0 = off.
#ifndef RT_STARVE_LIMIT
#define RT_STARVE_LIMIT 128 // config 2 at S = 16, same box: off 87.0 ms, 48 / 96 / 192 / 384 -> 80.6 / 80.0 / 79.8 / 80.1 (profiles/r03_ab_logs/r05w_split_ab.log); at S = 1 within noise
#endif
constexpr uint32_t kStarveLimit = RT_STARVE_LIMIT;
constexpr uint32_t kClaim = 64; // work items a wave claims per atomic
#ifndef RT_ACQUIRE_BATCH
#define RT_ACQUIRE_BATCH 1
#endif
constexpr uint32_t kAcquireBatch = RT_ACQUIRE_BATCH; // coarse schedule, tile claims: lanes out of work are served once this many wait (or nothing else is left to do)
#ifndef RT_DRAIN_LANES
#define RT_DRAIN_LANES 6
#endif
constexpr uint32_t kDrainLanes = RT_DRAIN_LANES; // fine schedule: a short phase runs once this many lanes wait for it
#ifndef RT_DRAIN_LANES_HEAVY
// (the long phases wait for 8: 1 M triangles, ms, at 12 node steps per vote: 3 lanes 987, 4: 959, 6: 928, 8: 924, 10: 940; 10 M triangles
// 1 011 / 998 / 987 / 984 / 998 -- profiles/r04au_heavy_drain_ab.log)
#define RT_DRAIN_LANES_HEAVY 8
#endif
constexpr uint32_t kDrainLanesHeavy = RT_DRAIN_LANES_HEAVY; // ... the same for the long phases (GEN, SHADE, LIGHT, SCATTER)
#ifndef RT_NODE_STEPS_PER_VOTE
#define RT_NODE_STEPS_PER_VOTE 12 // wide tree, 1 M triangles MIS at 1080p x 256 (ms), round 4, same box, interleaved: 8 steps 964, 10: 942, 11: 929,
                                  // 12: 928, 13: 932, 14: 936, 16: 949, 24: 1 020, 32: 1 108; 10 M triangles 984 at 12 - 13 against 990 at 16
                                  // (profiles/r04ar_fine_knobs_ab.log, r04as, r04at_node_steps_ab.log).  Drain thresholds 4 / 5 / 8 / 10 lanes
                                  // instead of 6: +2.5 / +0.7 / 0.0 / +2.7 %
#endif
constexpr int kNodeStepsPerVote = RT_NODE_STEPS_PER_VOTE;
#ifndef RT_FULL_WAVES
#define RT_FULL_WAVES 5 // waves per SIMD of the full-feature variants under the coarse schedule.  Round 2: 3 (a few dozen spilled
                        // registers, still faster than 2).  Round 4: the kernels need 119 - 125 VGPRs unconstrained (four 256-thread
                        // workgroups per CU); at a budget of 96 = five: all_materials naive 18.75 -> 17.36 ms, MIS 36.1 -> 36.5 (noise),
                        // same box (profiles/r04ae_small_ab.log)
#endif
#ifndef RT_FULL_FINE_WAVES
#define RT_FULL_FINE_WAVES 4 // full-feature variants under the fine schedule (200 k triangles + glass and GGX spheres:
                             // 2 waves 130 / 61 ms MIS / naive, 3 waves 114 / 48, 4 waves 114 / 42)
#endif
#ifndef RT_SIMPLE_FINE_WAVES
#define RT_SIMPLE_FINE_WAVES 4 // simple variants under the fine schedule: 128 VGPRs (about 40 spilled) against 168 at 3 waves:
                               // 100 k triangles naive 28.9 -> 25.2 ms, MIS 50.9 -> 48.8; 1 M: 37.7 -> 35.6, 62.6 -> 61.2
#endif
#ifndef RT_SIMPLE_COARSE_WAVES
#define RT_SIMPLE_COARSE_WAVES 5 // simple (triangles + lights) variants under the coarse schedule.  Round 2: 128 VGPRs with 20-40 spilled
                                 // registers beat 168 VGPRs at 3 waves (overshadowed MIS 212 -> 196 ms).  Round 4: the kernels need 93 - 121
                                 // VGPRs unconstrained; at a budget of 96 (FIVE waves per SIMD: five 256-thread workgroups per CU, one wave
                                 // per SIMD each -- the sky tables then stay in global memory) the config-3 kernel spills 39 registers and
                                 // is still 4.5 % faster, 114.4 -> 109.3 ms, same box (profiles/r04ad_simple_five_waves_ab.log); at 80 (six
                                 // waves, 124 spilled) 8.6 % slower (r04ab).  An extra wave hides more latency than the spills add.
#endif
#ifndef RT_SIMPLE_COARSE_BLOCK
#define RT_SIMPLE_COARSE_BLOCK 256
#endif
#ifndef RT_SPHERES_BLOCK
#define RT_SPHERES_BLOCK 512
#endif
#ifndef RT_FUSE_REGIONS
#define RT_FUSE_REGIONS(spheres_only) true // coarse schedule: which MIS kernels run a super-phase as ONE divergent region (see kFuseRegions)
#endif
#ifndef RT_BODY_ALWAYS
#define RT_BODY_ALWAYS(spheres_only) true // coarse schedule: which kernels run the loop body without the `nothing to run` arm (see there)
#endif
#ifndef RT_SPHERES_MAX_BLOCK
#define RT_SPHERES_MAX_BLOCK 768 // coarse spheres-only kernels (FeatPair among them): the host launches two workgroups of 512 OR 768 threads per
                                 // CU, whichever puts more waves on it at the kernel's register count (rt_api.cpp; 66 - 100 VGPRs)
#endif
#ifndef RT_PAIR_WAVES
#define RT_PAIR_WAVES 6 // the two-sphere special case (FeatPair): 75 VGPRs (MIS), 66 (naive).  Round 4, config 2, same box: 512 x 3 without the
                        // sky tables in LDS 62.0 ms, 640 x 2 (five waves per SIMD) 62.5, 768 x 2 (six) 61.2
#endif
#ifndef RT_SPHERES_WAVES
#define RT_SPHERES_WAVES 4 // waves per SIMD the spheres-only variants are register-limited to under the FINE schedule (119 - 128 VGPRs)
#endif
#ifndef RT_SPHERES_COARSE_WAVES
// ... and under the coarse schedule: a budget of 80 VGPRs = SIX waves per SIMD.  The exhaustive MIS kernel needs 93 without the
// limit and spills 6 registers (28 bytes of scratch) with it: config 2 through the general kernel 69.0 -> 62.1 ms, same box
// (profiles/r04aa_spheres_six_waves_ab.log) -- two more waves per SIMD are worth 10 % to an issue-bound kernel, far more than the
// issue microbenchmark's 2.5 -> 2.36 cycles: they also hide each other's latencies.  (The config-3 kernel, 121 VGPRs, spills 124
// at that budget: 114.4 -> 124.2 ms, profiles/r04ab_simple_six_waves_ab.log; at 96, five waves, it gains: RT_SIMPLE_COARSE_WAVES.)
#define RT_SPHERES_COARSE_WAVES 6
#endif

__device__ __forceinline__ float power_heuristic(float pdf_a, float pdf_b) // rt_core/src/lib.rs:36-40
{
	const float a_sq = pdf_a * pdf_a;
	return a_sq / (a_sq + pdf_b * pdf_b);
}

// work item -> pixel.  Work items enumerate this shard's tiles (tile t belongs to shard
// t % shard_count) row-major inside each tile, so the 64 lanes of a wave start on one compact
// tile_w x tile_h block of the image (coherent primary rays).
// (Handing tiles out bottom-up, so that the tail of the render is cheap sky pixels, was measured
// SLOWER on rtweekend1: 181 ms vs 154 ms.  Waves that mix cheap and expensive pixels vote better.)
__device__ __forceinline__ bool work_to_pixel(const DevRenderParams &P, uint32_t w, uint32_t &x, uint32_t &y)
{
	const uint32_t tile_pixels = P.tile_w * P.tile_h;
	const uint32_t k = w / tile_pixels;
	const uint32_t in = w - k * tile_pixels;
	const uint32_t tile = P.shard_index + k * P.shard_count;
	const uint32_t ty = tile / P.tiles_x;
	const uint32_t tx = tile - ty * P.tiles_x;
	x = tx * P.tile_w + in % P.tile_w;
	y = ty * P.tile_h + in / P.tile_w;
	return x < P.width && y < P.height;
}

// Workgroup size and register budget per feature set (the macros above say what each was measured against).  Coarse schedule:
// spheres-only kernels (FeatPair among them) are declared for six waves per SIMD and start from 512 threads -- the host launches
// two workgroups of 512 or 768 per CU, whichever puts more waves on it (rt_api.cpp), 768 x 2 keeps the 54 KB sky tables in LDS;
// triangles + lights and full-feature kernels are declared for five and run as five 256-thread workgroups (one wave per SIMD each;
// the sky tables then stay in global memory).  Fine-schedule kernels keep 256 threads (their LDS goes to the traversal stacks) at
// four waves per SIMD.
template <class F, bool FINE = false, bool XCHG = false> struct KernelShape {
	static constexpr bool spheres_only = !(F::tri || F::lights || F::cmat || F::ctex);
	static constexpr bool simple = !spheres_only && !(F::cmat || F::ctex);
	// (the exchange variants -- opt-in, measured slower, kept for their tests -- keep the register budgets they were built and measured with)
	static constexpr int waves_per_simd = (F::cmat || F::ctex) ? (FINE ? RT_FULL_FINE_WAVES : (XCHG ? 3 : RT_FULL_WAVES))
	                                      : (F::pair ? RT_PAIR_WAVES : (spheres_only ? ((FINE || XCHG) ? RT_SPHERES_WAVES : RT_SPHERES_COARSE_WAVES)
	                                                                                 : (FINE ? RT_SIMPLE_FINE_WAVES : (XCHG ? 4 : RT_SIMPLE_COARSE_WAVES))));
	// (fine schedule with the exchange: eight waves share the pools of walk starts and walk results, one of them shades)
	static constexpr int block = (FINE && XCHG) ? 512 : (spheres_only ? RT_SPHERES_BLOCK : ((simple && !FINE) ? RT_SIMPLE_COARSE_BLOCK : 256));
	// the largest workgroup the kernel may be launched with (its __launch_bounds__); `block` is what the host starts from
	static constexpr int max_block = ((spheres_only || simple) && !FINE && !XCHG && block == 512) ? RT_SPHERES_MAX_BLOCK : block;
};

#ifdef RT_STATS
// section timing of the coarse schedule (diagnostic build): wave wall-clock cycles per section, lane 0
#define RT_SECTION(k) do { const unsigned long long now_ = wall_clock64(); st_sect[(k)] += now_ - st_mark; st_mark = now_; } while (0)
#else
#define RT_SECTION(k) do { } while (0)
#endif
#ifdef RT_STATS
// diagnostic build only: schedule statistics, read back with hipMemcpyFromSymbol by tests/probes/gpu_stats_probe.py
__device__ unsigned long long g_stats[64];
__device__ unsigned long long g_hist[4][65]; // coarse schedule, by lane count: [0] PRIMARY iterations by participating lanes, [1] by
                                             // lanes waiting for BOUNCE at that moment, [2] acquire events by needy lanes, [3] BOUNCE iterations
#endif

// Cross-wave exchange of path state (XCHG, coarse MIS kernels, opt-in through RT_TUNE_EXCHANGE): a lane whose path
// waits for the super-phase its wave is NOT about to run parks its state in a workgroup pool in LDS and takes a parked
// state of the kind the wave does run, so both super-phases see full waves.  A path's whole state travels (40 dwords for
// a path between bounces, 10 for a pixel between samples), the pixel with it, so a pixel's samples are still folded in
// order and every pixel keeps its own random streams: which lane finishes a pixel cannot change it.  The pool is
// guarded by ONE try-lock: a wave that does not get it skips the exchange for this iteration and votes as usual --
// nothing ever waits, so nothing can hang.
// Records are read and written 16 bytes at a time; the strides (in dwords) are = 4 * odd mod 64... chosen so that the 16 lanes
// one ds_*_b128 pass serves land on 16 different groups of four banks: 44 = 4 * 11, 12 = 4 * 3.
constexpr uint32_t kXchgBStride = 44, kXchgPStride = 12; // dwords per parked path (40 used) / pixel (10 used)
constexpr uint32_t kXchgMaxSlots = 64;                   // parked states of each kind per workgroup, at most (DevRenderParams.xchg_slots)
constexpr uint32_t kXchgMinLanes = 4;                    // fewer idle lanes than this are not worth the trade
// Fine schedule (big trees): the short phases (GEN / SHADE / LIGHT / SCATTER) run on a handful of lanes because most
// lanes of a wave are in the middle of a walk, and every execution costs the wave the same whatever the lane count.
// With XCHG the waves of a workgroup take roles: wave 0 SHADES, the others WALK.  A walker parks the lanes whose walk
// just ended (pending SHADE or SCATTER) in pool R and refills them from pool W with walks waiting to start; the shader
// takes walk results from R by the wavefront, runs the short phases on full waves, parks the walks they start in W.  A
// record is the lane's whole state (60 dwords: path, pixel, random stream, ray, walk result, light sample); walks in
// progress never move (their stack stays where it is).  A slot is claimed under the workgroup's lock (a few dozen cycles:
// one state word per slot, one lane per slot), copied outside it and published by its state word, so waves do not queue
// behind each other's copies.  Nothing ever waits: a wave that cannot trade (lock taken, pool full or empty, the other
// role gone) carries on with the per-wave schedule, which remains complete on its own.
constexpr uint32_t kXchgRecWords = 60;        // 15 x 16 bytes; = 4 * 15: sixteen lanes of a b128 pass hit sixteen bank groups
constexpr uint32_t kXchgFineSlots = 64;       // records per pool: one lane looks after one slot
constexpr uint32_t kXchgFineHdr = 8 + 2 * 64; // lock, counts, alive flags; one state word per slot of W and of R
#ifndef RT_XCHG_SHADERS
#define RT_XCHG_SHADERS 1
#endif
constexpr uint32_t kXchgShaderWaves = RT_XCHG_SHADERS; // waves of a workgroup (the first ones) that shade
enum : uint32_t { XS_EMPTY = 0, XS_FULL = 1, XS_BUSY = 2 };

// Everything a render launch is told, as ONE by-value kernel argument, so that it sits at offset 0 of the kernarg segment and
// the kernel can read any field from there with a scalar load WHERE IT IS USED.  Taken as ordinary by-value parameters, the
// compiler loads all of them in the prologue and keeps them in SGPRs for the life of the persistent loop: 170 dwords of
// arguments against 100 usable SGPRs meant 65 (spheres-only) to 105 (simple variants) of them lived in VGPR lanes and
// came back through v_readlane / v_writelane inside the loop.  Fields a phase needs on every execution and that are
// few (max_depth, rr_threshold, the scene's tables) stay ordinary values; the camera, the seed, the sample window and
// the tile geometry (used once per sample or once per pixel) are re-read through kargs() below.
struct RenderArgs {
	DevScene S;
	DevCamera cam;
	DevRenderParams P;
	float *out;
	unsigned long long *rays_shot;
	uint32_t *work_counter;
	uint32_t *stack_ovf;
	DevPairScene pair; // FeatPair only (rt_types.h): the whole scene, read where it is used
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef const __attribute__((address_space(4))) RenderArgs *KArgs; // the kernarg segment is constant memory: s_load
#else
typedef const RenderArgs *KArgs; // (host pass: the kernel body is only parsed)
#endif

template <int METHOD, bool PRUNE, bool FINE, bool SKY_LDS, class F, bool XCHG = false>
__global__ __launch_bounds__((KernelShape<F, FINE, XCHG>::max_block), (KernelShape<F, FINE, XCHG>::waves_per_simd)) void render_kernel(const RenderArgs args_by_value)
{
	extern __shared__ __align__(16) uint32_t lds[];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;

	(void)args_by_value; // read through the kernarg pointer only
#if defined(__HIP_DEVICE_COMPILE__)
	const KArgs K = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
	// a pointer the optimiser cannot see through: loads made through it stay where they are written (no hoisting out of the loop)
	auto kargs = [&]() -> KArgs {
		KArgs k = K;
		asm volatile("" : "+s"(k));
		return k;
	};
#else
	const KArgs K = &args_by_value;
	auto kargs = [&]() -> KArgs { return K; };
#endif
	// FeatPair: the scene in scalar registers for the length of a super-phase -- read from the kernel arguments in ONE round of
	// scalar loads at its top (rt_types.h DevPairScene), by the walks, make_hit and the material evaluations after that
	DevPairScene PS = {};
	auto load_pair_scene = [&]() {
		if constexpr (F::pair) {
			const KArgs k = kargs();
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				PS.c0min[i] = k->pair.c0min[i]; PS.c0max[i] = k->pair.c0max[i];
				PS.c1min[i] = k->pair.c1min[i]; PS.c1max[i] = k->pair.c1max[i];
				PS.sky_c1[i] = k->pair.sky_c1[i]; PS.sky_c2[i] = k->pair.sky_c2[i];
			}
			PS.slot0 = k->pair.slot0; PS.slot1 = k->pair.slot1; PS.rank0 = k->pair.rank0; PS.rank1 = k->pair.rank1;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				PS.sphere[0][i] = k->pair.sphere[0][i]; PS.sphere[1][i] = k->pair.sphere[1][i];
				PS.lambert[0][i] = k->pair.lambert[0][i]; PS.lambert[1][i] = k->pair.lambert[1][i];
			}
			PS.inv_radius[0] = k->pair.inv_radius[0]; PS.inv_radius[1] = k->pair.inv_radius[1];
			PS.sky_param = k->pair.sky_param;
			PS.sky_tex_type = k->pair.sky_tex_type;
		}
	};
	// (Passing the node / primitive / rank arrays of the two-leaf walk as separate `const __restrict__` kernel arguments turns
	// its loads back into scalar loads -- through this struct they are vector loads of a uniform address -- but measured nothing
	// on configs 2 and 3 and cost the fine MIS kernels 10 % (1 M triangles 50.1 -> 55.5 ms, same-box A/B): three more
	// arguments shifted their register allocation.)
	const DevScene S_global = K->S;
	const DevRenderParams P = K->P; // (fields used only through kargs() are never loaded from this copy)
	uint32_t *const stack_ovf = K->stack_ovf;
	DevScene S = S_global;

	// ---- stage the sky CDF rows + marginal CDF into LDS ----
	SkyTables T;
	uint32_t sky_words = 0;
	if (SKY_LDS) {
		const uint32_t n_rows = S.sky.res_y * (S.sky.res_x + 1u);
		const uint32_t n_all = n_rows + S.sky.res_y + 1u;
		float *lds_sky = reinterpret_cast<float *>(lds);
		for (uint32_t i = threadIdx.x; i < n_all; i += blockDim.x)
			lds_sky[i] = S.sky.row_cdf[i]; // marginal follows the rows in the same allocation
		__syncthreads();
		sky_words = (n_all + 3u) & ~3u;
		// ... and the guide tables behind them (a multiple of 16 bytes: guide_k is a power of two >= 16)
		const uint32_t guide_words = (S.sky.res_y + 1u) * S.sky.guide_k / 4u;
		uint32_t *lds_guide = lds + sky_words;
		const uint32_t *src_guide = reinterpret_cast<const uint32_t *>(S.sky.guide);
		for (uint32_t i = threadIdx.x; i < guide_words; i += blockDim.x)
			lds_guide[i] = src_guide[i];
		__syncthreads();
		T.row_cdf = lds_sky;
		T.marginal_cdf = lds_sky + n_rows;
		T.guide = reinterpret_cast<const uint8_t *>(lds_guide);
		sky_words += guide_words;
	} else {
		T.row_cdf = S.sky.row_cdf;
		T.marginal_cdf = S.sky.marginal_cdf;
		T.guide = S.sky.guide;
	}
	T.guide_k = S.sky.guide_k;
	T.inv_res = reinterpret_cast<KWords>(&K->S.sky.inv_res_ok);
	// ---- tiny scenes (coarse schedule only): stage the whole scene into LDS, so the dependent loads
	// of a walk (node -> primitive -> material -> texture) pay LDS latency instead of L1/L2 latency ----
	uint32_t blob_words = 0;
	if (!FINE && P.scene_in_lds) {
		uint32_t *lds_blob = lds + sky_words;
		blob_words = S.blob_bytes / 4u;
		for (uint32_t i = threadIdx.x; i < blob_words; i += blockDim.x)
			lds_blob[i] = S.blob[i];
		__syncthreads();
		const char *b = reinterpret_cast<const char *>(lds_blob);
		S.nodes = reinterpret_cast<const DevNode *>(b + S.off_nodes);
		S.prims = reinterpret_cast<const DevPrim *>(b + S.off_prims);
		S.shade = reinterpret_cast<const DevShade *>(b + S.off_shade);
		S.prim_rank = reinterpret_cast<const uint32_t *>(b + S.off_rank);
		S.materials = reinterpret_cast<const DevMaterial *>(b + S.off_materials);
		S.textures = reinterpret_cast<const DevTexture *>(b + S.off_textures);
		S.lights = reinterpret_cast<const uint32_t *>(b + S.off_lights);
		S.big_leaves = reinterpret_cast<const uint2 *>(b + S.off_big_leaves);
	}
	StackMem SM;
	SM.cap = P.stack_cap;
	SM.ovf_depth = P.stack_ovf_depth;
	SM.ovf = stack_ovf;
	SM.region = lds + sky_words + blob_words;
	uint32_t *stk = SM.region + wave * (P.stack_cap * kStackStride) + lane;
	// exchange pool (XCHG): [lock, parked paths, parked pixels, pad] [paths: word k of slot s at k * slots + s] [pixels likewise]
	// (computed under `if constexpr` on purpose: as a dead expression in the kernels without the exchange it still cost the
	// fine-schedule kernels 12 more spilled registers and 3.5 % on 1 M triangles)
	uint32_t *pool = nullptr;
	if constexpr (XCHG) {
		pool = lds + (((uint32_t)(SM.region - lds) + (blockDim.x >> 6) * (P.stack_cap * kStackStride) + 3u) & ~3u);
		if (FINE) { // [lock, records in W, records in R, shaders alive, walkers alive, -, -, -] [state of W's slots] [of R's]
			if (threadIdx.x < kXchgFineHdr)
				pool[threadIdx.x] = threadIdx.x == 3u ? kXchgShaderWaves : (threadIdx.x == 4u ? (blockDim.x >> 6) - kXchgShaderWaves : 0u);
		} else if (threadIdx.x < 4u) {
			pool[threadIdx.x] = 0u;
		}
		__syncthreads();
	}

#ifdef RT_STATS
	unsigned long long st_iters[2] = {0, 0}, st_active[2] = {0, 0}, st_gen = 0, st_hist[8] = {};
	// [0..8]: the sections of the coarse loop; [9..15]: their parts (tests/probes/gpu_stats_probe.py names them)
	unsigned long long st_sect[16] = {}, st_mark = wall_clock64();
	unsigned long long st_fine_clock[PH_COUNT + 1] = {};
	unsigned long long st_fine_iters[PH_COUNT] = {}, st_fine_active[PH_COUNT] = {};
#endif
	const bool sky_samplable = sky_can_sample(S);
	constexpr int known_path = F::known_materials ? kMatLambertian : kMatRead; // the material a path continues from (see do_shade)

	// ---- per-lane state (registers) ----
	int ph = PH_NEED_PIXEL;
	rt_rng rng = {1u, 2u, 3u, 4u};
	// the walk in progress: its ray, where it stands, what it has found
	Ray ray;
	ray.o = ray.d = ray.inv = ray.shear = v3s(0.0f);
	uint32_t node = kRefDone;
	int sp = 0;
	float best_t = 0.0f;          // closest walk: best t so far
	uint32_t best_prim = kNoPrim; // closest walk: best primitive; any-hit walk: kNoPrim until occluded
	bool any_hit = false;         // the walk is a shadow walk
	// the path
	V3 thr = v3s(1.0f), outp = v3s(0.0f), mean = v3s(0.0f), wo = v3s(0.0f);
	Hit hit;
	hit.t = 0.0f;
	hit.point = hit.error = hit.normal = v3s(0.0f);
	hit.err_dot = 0.0f;
	hit.uvx = hit.uvy = 0.0f;
	hit.has_uv = hit.out = false;
	uint32_t mat = 0;
	uint32_t depth = 0, sample_local = 0, out_index = 0, pixel_index = 0, px = 0, py = 0;
	uint32_t chunk_begin = 0, chunk_n = P.spp; // the passes [chunk_begin, chunk_begin + chunk_n) this lane folds for its pixel
	uint32_t ray_count = 0;
	unsigned long long rays_total = 0;
	bool primary = true;
	// the light sample in flight (LIGHT -> shadow walk -> SCATTER).  With the fine schedule it
	// survives loop iterations (PL below, and the shadow ray lives in `ray`); with the coarse schedule
	// LIGHT, the shadow walk and SCATTER run back to back in one iteration, so the context and the
	// shadow ray are locals of that iteration and the registers are free the rest of the time.
	struct LightCtx {
		V3 l_wi;
		float pdf_multiplier;
		float t_limit;  // occluders need NOT (t >= t_limit); NaN = no limit (sky)
		uint32_t skip;  // the light primitive itself (check_hit_index skips it)
		bool have_shadow, shadow_is_sky;
	};
	LightCtx PL;
	PL.l_wi = v3s(0.0f);
	PL.pdf_multiplier = 1.0f;
	PL.t_limit = 0.0f;
	PL.skip = kNoPrim;
	PL.have_shadow = PL.shadow_is_sky = false;

	// start a walk of `ray`: the root test of Bvh::get_intersection_candidates (mod.rs:203-210)
	auto begin_walk = [&](bool shadow) {
		any_hit = shadow;
		best_t = 0.0f;
		best_prim = kNoPrim;
		sp = 0;
		if (!FINE) { // coarse schedule: the whole walk runs inside the super-phase (walk_pending below)
			ph = PH_NODE;
			return;
		}
		if (XCHG) { // the walk may start in another wave: start_walk below, after the exchange
			ph = PH_WALK;
			return;
		}
		if (root_box_misses(S, ray)) {
			node = kRefDone;
			ph = shadow ? PH_SCATTER : PH_SHADE;
		} else if (S.narrow_only != 0u || !ray_is_regular(ray)) {
			ph = PH_NARROW; // a zero direction component or a non-finite origin: only the two-child walk is exact for it
		} else {
			node = S.root4_ref; // the fine schedule walks the wide tree (the host selects it only when one exists)
			ph = ref_is_leaf(node) ? PH_LEAF : PH_NODE;
		}
	};
	// PH_WALK -> the first step of the walk (what begin_walk does at once without the exchange)
	auto start_walk = [&]() {
		sp = 0;
		if (root_box_misses(S, ray)) {
			node = kRefDone;
			ph = any_hit ? PH_SCATTER : PH_SHADE;
		} else if (S.narrow_only != 0u || !ray_is_regular(ray)) {
			ph = PH_NARROW;
		} else {
			node = S.root4_ref;
			ph = ref_is_leaf(node) ? PH_LEAF : PH_NODE;
		}
	};
	// the walk moved to `node` (inner, leaf or finished): choose the lane's next phase
	auto after_step = [&]() {
		if (node == kRefDone)
			ph = any_hit ? PH_SCATTER : PH_SHADE;
		else
			ph = ref_is_leaf(node) ? PH_LEAF : PH_NODE;
	};
	// sample finished: filter, fold into the running mean, next sample or next pixel
	auto finalize = [&](bool filter) {
		V3 c = outp;
		if (filter && (contains_nan(c) || !is_finite_any(c))) // integrators/mod.rs:74-76, mis.rs:88-90
			c = v3s(0.0f);
		if (P.sample_split > 1u) { // (wave-uniform) a chunk of a pixel's passes: their SUM, in pass order (rt_hip.h sample_split)
			mean.x += c.x;
			mean.y += c.y;
			mean.z += c.z;
		} else {
			const float i_f = (float)(sample_local + 1u);
			mean.x += (c.x - mean.x) / i_f; // src/main.rs:179-185
			mean.y += (c.y - mean.y) / i_f;
			mean.z += (c.z - mean.z) / i_f;
		}
		rays_total += ray_count;
		sample_local += 1;
		if (sample_local == chunk_n) {
			float *const out = kargs()->out;
			out[3u * (size_t)out_index + 0u] = mean.x;
			out[3u * (size_t)out_index + 1u] = mean.y;
			out[3u * (size_t)out_index + 2u] = mean.z;
			ph = PH_NEED_PIXEL;
		} else {
			ph = PH_GEN;
		}
	};

	// GEN -- new sample: the pixel loop body of sample_image, random_sampler.rs:50-61
	auto do_gen = [&]() {
		const KArgs k = kargs(); // seed, sample window, image size and camera: scalar loads here, once per sample
		const uint64_t seed = ((uint64_t)k->P.seed_hi << 32) | k->P.seed_lo;
		const uint64_t sample_begin = ((uint64_t)k->P.sample_begin_hi << 32) | k->P.sample_begin_lo;
		const V3 cam_o = v3(k->cam.origin[0], k->cam.origin[1], k->cam.origin[2]);
		const V3 cam_ll = v3(k->cam.lower_left[0], k->cam.lower_left[1], k->cam.lower_left[2]);
		const V3 cam_h = v3(k->cam.horizontal[0], k->cam.horizontal[1], k->cam.horizontal[2]);
		const V3 cam_v = v3(k->cam.vertical[0], k->cam.vertical[1], k->cam.vertical[2]);
		// (the image size in the same round of scalar loads as the rest: left to the compiler it is fetched after the Philox rounds,
		// a round trip of its own)
		const uint32_t w1 = here_(k->P.width - 1u), h1 = here_(k->P.height - 1u);
		rt_rng_seed(&rng, seed, (uint64_t)pixel_index, sample_begin + chunk_begin + sample_local);
		RT_SECTION(15); // GEN: loads + stream seed
		// (jitter + pixel) / (W - 1): the numerator is zero or in [2^-23, 2^31), the denominator in [1, 2^31]: tame (rt_lean.h) --
		// and when the host has verified the two divisors (DevRenderParams::w1h1_ok), two fma steps on its reciprocals
		const float jx = rt_rng_range_f32(&rng, 0.0f, 1.0f) + (float)px, jy = rt_rng_range_f32(&rng, 0.0f, 1.0f) + (float)py;
		float u, v;
		if (here_(k->P.w1h1_ok) != 0u) { // (wave-uniform)
			u = div_by_verified(jx, (float)w1, k->P.inv_w1);
			v = 1.0f - div_by_verified(jy, (float)h1, k->P.inv_h1);
		} else {
			u = div_tame_fix_(jx, (float)w1);
			v = 1.0f - div_tame_fix_(jy, (float)h1);
		}
		// SimpleCamera::get_ray  camera.rs:57-63 (draws an unused `time`)
		ray = ray_new<F>(cam_o, cam_ll + cam_h * u + cam_v * v - cam_o);
		(void)rt_rng_f32(&rng);
		thr = v3s(1.0f);
		outp = v3s(0.0f);
		depth = 0;
		ray_count = 0;
		primary = true;
		begin_walk(false);
	
	};

#ifdef RT_STATS
	unsigned long long st_lane_nodes = 0, st_lane_prims = 0, st_lane_maxsp = 0;
#endif
	// NODE -- one inner-node step: Bvh::get_intersection_candidates' loop body (mod.rs:203-221)
	auto do_node = [&]() {
#ifdef RT_STATS
		st_lane_nodes += 1;
		if ((unsigned long long)sp > st_lane_maxsp)
			st_lane_maxsp = (unsigned long long)sp;
#endif
		const bool limit_valid = any_hit ? !(PL.t_limit != PL.t_limit) : (best_prim != kNoPrim);
		// wave-uniform: trees whose worst case fits the LDS columns (all but the deepest) take the walk without capacity checks
		if (SM.ovf_depth == 0u)
			node = descend4<PRUNE, false>(S, SM, ray, node, stk, sp, limit_valid, any_hit ? PL.t_limit : best_t);
		else
			node = descend4<PRUNE, true>(S, SM, ray, node, stk, sp, limit_valid, any_hit ? PL.t_limit : best_t);
		after_step();
	
	};

	// LEAF -- one leaf: the primitive loops of Bvh::check_hit (mod.rs:270-293) / check_hit_index (:244-261)
	auto do_leaf = [&]() {
		uint32_t first, count, leaf_ref;
		// the walk got here through conservative boxes: the leaf is a candidate iff its exact box passes the
		// reference's test (rt_intersect.h, the wide walk); the box record also says which primitives the leaf holds
		const bool candidate = wide_leaf_hit(S, node, ray, leaf_ref);
		leaf_range(S, leaf_ref, first, count);
		if (!candidate)
			count = 0u;
		bool occluded = false;
		for (uint32_t slot = first; slot < first + count; ++slot) {
			if (any_hit && slot == PL.skip)
				continue;
#ifdef RT_STATS
			st_lane_prims += 1;
#endif
			const PrimGeom g = load_prim<F>(S, slot);
			float t;
			if (prim_t<F>(g, ray, t) && t > 0.0f) {
				if (any_hit) {
					if (!(t >= PL.t_limit)) {
						occluded = true;
						break;
					}
				} else {
					bool take;
					if (best_prim == kNoPrim)
						take = true;
					else if (t < best_t)
						take = true;
					else if (t == best_t)
						take = S.prim_rank[slot] < S.prim_rank[best_prim]; // first in BFS-leaf order wins
					else
						take = false;
					if (take) {
						best_t = t;
						best_prim = slot;
					}
				}
			}
		}
		if (occluded) {
			best_prim = 0u; // any-hit result: anything but kNoPrim means "occluded"
			node = kRefDone;
		} else if (sp == 0) {
			node = kRefDone;
		} else {
			--sp;
			node = SM.ovf_depth == 0u ? stack_load<false>(SM, stk, sp) : stack_load<true>(SM, stk, sp);
		}
		after_step();
	
	};

	// SHADE -- the walk of a path ray ended: integrators/mod.rs:31-72 (naive), mis.rs:17-33,50-86 (MIS)
	// `arm`: 0 = either arm of the MIS shade (fine schedule), 1 = only primary lanes can be here, 2 = only
	// bounce lanes can be here (coarse schedule, see the super-phases below): the other arm is not compiled in
	auto do_shade = [&](auto arm_tag) {
		constexpr int arm = decltype(arm_tag)::value;
		const uint32_t prim = best_prim;
		bool finish = false;
		bool filter = true;
		Hit nh;
		uint32_t nmat;
		if (prim != kNoPrim)
			make_hit<F>(S, prim, ray, best_t, nh, nmat, PS);
		else
			make_sky_hit_lean<F>(S, nh, nmat);
		// what is known of the two materials in play without reading them (rt_shade.h, kMatRead): the new hit's follows from
		// what was hit; a path only ever continues from a material that is not a light
		const int known_new = F::known_materials ? (prim == kNoPrim ? kMatEmit : kMatLambertian) : kMatRead;

		if (METHOD == 0) {
			// ---- NaiveIntegrator::get_colour loop body  integrators/mod.rs:31-72 ----
			ray_count += 1;
			const V3 wo_n = ray.d;
			const V3 emission = emission_of_hit<F>(S, PS, nmat, prim == kNoPrim, nh, wo_n);
			const bool exit = mat_scatter_ray<F>(S, nmat, ray, nh, rng, known_new);
			if (depth == 0) {
				outp = outp + emission;
				if (exit)
					finish = true;
			}
			if (!finish && exit) {
				outp = outp + thr * emission;
				finish = true;
			}
			if (!finish) {
				if (!mat_is_delta<F>(S, nmat))
					thr = thr * mat_eval_over_pdf<F>(S, nmat, nh, wo_n, ray.d, known_new, PS);
				else
					thr = thr * mat_eval<F>(S, nmat, nh, wo_n, ray.d, known_new, PS);
				if (depth > P.rr_threshold) {
					const float p = component_max(thr);
					if (rt_rng_f32(&rng) > p)
						finish = true;
					else
						thr = thr / p;
				}
				if (!finish) {
					depth += 1;
					if (!(depth < P.max_depth))
						finish = true;
				}
			}
		} else if (arm == 1 || (arm == 0 && primary)) {
			// ---- MisIntegrator::get_colour prologue  mis.rs:17-33 ----
			wo = ray.d;
			hit = nh;
			mat = nmat;
			RT_SECTION(13); // primary arm: hit record
			const V3 emission = emission_of_hit<F>(S, PS, mat, prim == kNoPrim, hit, wo);
			Ray clone = ray; // scatter on a clone: draws consumed, ray discarded (mis.rs:25)
			const bool exit = mat_scatter_ray<F>(S, mat, clone, hit, rng, known_new);
			outp = outp + emission;
			RT_SECTION(14); // primary arm: emission + scatter on a clone
			if (exit) {
				finish = true;
				filter = false; // mis.rs:29-31 returns before the filter
			} else {
				depth = 1;
				primary = false;
				if (!(depth < P.max_depth))
					finish = true;
			}
		} else {
			// ---- material-sampling half of the MIS loop  mis.rs:50-86 ----
			const V3 m_wi = ray.d;
			const float m_pdf = mat_scattering_pdf<F>(S, mat, hit, wo, m_wi, known_path);
			const V3 le = emission_of_hit<F>(S, PS, nmat, prim == kNoPrim, hit /* the OLD hit, mis.rs:55 */, m_wi);
			thr = thr * mat_eval_over_pdf<F>(S, mat, hit, wo, m_wi, known_path, PS);
			RT_SECTION(11); // bounce arm: hit record, pdf, emission, throughput
			if (!is_zero(le)) {
				// bvh.get_samplable().contains(&index): Bvh.lights is exactly the primitives whose material
				// is_light() (acceleration/mod.rs:84-88), so the hit primitive's material answers it
				const bool on_light = F::lights && (prim != kNoPrim) && mat_is_light(S, nmat) && !mat_is_delta<F>(S, mat);
				if (on_light || (prim == kNoPrim && sky_samplable)) {
					// Bvh::get_pdf_from_index  acceleration/mod.rs:299-318
					const uint32_t n_l = F::lights ? S.n_lights : 0u;
					const uint32_t n_choices = sky_samplable ? n_l + 1u : n_l;
					float l_pdf;
					if (prim == kNoPrim) {
						l_pdf = div_by_count(sky_pdf(S, T, m_wi), n_choices);
					} else {
						const PrimGeom g = load_prim<F>(S, prim);
						l_pdf = div_by_count(prim_scattering_pdf<F>(g, hit.point, m_wi, nh), n_choices);
					}
					const float mis_weight = power_heuristic(m_pdf, l_pdf);
					outp = outp + thr * le * mis_weight;
				} else {
					outp = outp + thr * le;
				}
			}
			RT_SECTION(12); // bounce arm: MIS weight of what was hit (sky_pdf)
			if (mat_is_light<F>(S, nmat, known_new)) {
				finish = true;
			} else {
				if (depth > P.rr_threshold) {
					const float p = component_max(thr);
					if (rt_rng_f32(&rng) > p)
						finish = true;
					else
						thr = thr / p;
				}
				if (!finish) {
					wo = m_wi;
					hit = nh;
					mat = nmat;
					depth += 1;
					if (!(depth < P.max_depth))
						finish = true;
				}
			}
		}
		if (finish)
			finalize(filter);
		else if (METHOD == 1)
			ph = PH_LIGHT; // the path continues: light sampling is next
		else
			begin_walk(false); // naive: `ray` already is the scattered ray
	
	};

	// LIGHT -- sample_lights up to the shadow ray  mis.rs:95-157
	auto do_light = [&](LightCtx &L, Ray &sr) {
		ray_count += 1; // mis.rs:38
		L.have_shadow = false;
		L.shadow_is_sky = false;
		L.pdf_multiplier = 1.0f;
		L.skip = kNoPrim;
		const uint32_t samplable_len = F::lights ? S.n_lights : 0u;
		bool pick_sky = false, pick_light = false;
		uint32_t light_slot = 0;
		if (samplable_len == 0u) {
			pick_sky = sky_samplable; // (0,true) => sample_sky(1.0); (0,false) => None
		} else if (!sky_samplable) {
			L.pdf_multiplier = 1.0f / (float)samplable_len;
			light_slot = rt_rng_below(&rng, samplable_len); // gen_range(0..len)
			pick_light = true;
		} else {
			L.pdf_multiplier = 1.0f / (float)(samplable_len + 1u);
			light_slot = rt_rng_below(&rng, samplable_len + 1u); // gen_range(0..=len)
			if (light_slot == samplable_len)
				pick_sky = true;
			else
				pick_light = true;
		}
		const V3 shadow_origin = hit.point + 0.0001f * hit.normal;
		// the two arms only choose the direction; the shadow ray itself (Ray::new: some fifty instructions) is formed once, after
		// them, for the lanes of both -- inside each arm the wave paid for it twice whenever its lanes disagreed on the pick
		PrimGeom g;
		g.type = g.material = 0u;
		g.p0 = g.p1 = g.p2 = v3s(0.0f);
		if (pick_sky) {
			L.l_wi = sky_sample(S, T, rng);
			RT_SECTION(9); // LIGHT: sky_sample
			L.t_limit = __uint_as_float(0x7FC00000u); // NaN: any t > 0 occludes
			L.have_shadow = true;
			L.shadow_is_sky = true;
		} else if (pick_light) {
			if (S.single_light != kNoPrim) { // (wave-uniform) the one light: its record through scalar loads, from the scene in global memory
				L.skip = S.single_light;
				g = load_prim_uniform<F>(&S_global.prims[L.skip]);
			} else {
				L.skip = S.lights[light_slot];
				g = load_prim<F>(S, L.skip);
			}
			L.l_wi = prim_sample_visible_from_point<F>(g, hit.point, rng);
		}
		if (pick_sky || pick_light)
			sr = ray_new<F>(shadow_origin, L.l_wi);
		if (pick_light) {
			float lt;
			if (prim_t<F>(g, sr, lt) && lt > 0.0f) { // Bvh::check_hit_index  mod.rs:231-242
				L.t_limit = lt;
				L.have_shadow = true;
			}
		}
		if (L.have_shadow) {
			begin_walk(true);
		} else {
			best_prim = kNoPrim;
			ph = PH_SCATTER;
		}
	
	};

	// SCATTER -- the shadow walk ended: light contribution (mis.rs:39-43) and material sampling (mis.rs:46-49)
	auto do_scatter = [&](LightCtx &L, Ray &sr) {
		const bool occluded = best_prim != kNoPrim;
		if (L.have_shadow && !occluded) {
			bool valid = false;
			V3 le = v3s(0.0f);
			float l_pdf = 0.0f;
			if (L.shadow_is_sky) { // sample_sky  mis.rs:104-115
				le = emission_of_hit<F>(S, PS, S.sky.material, true, hit, L.l_wi);
				l_pdf = sky_pdf(S, T, L.l_wi) * L.pdf_multiplier;
				valid = true;
			} else { // sample_light  mis.rs:117-133 (`ray` still is the shadow ray)
				Hit lh;
				uint32_t lm;
				PrimGeom g;
				if (S.single_light != kNoPrim) { // (wave-uniform) as in do_light; a sphere's hit record needs nothing but this record
					g = load_prim_uniform<F>(&S_global.prims[L.skip]);
					if (!F::tri || g.type == kPrimSphere) {
						make_sphere_hit_by_reciprocal(g.p0, g.p1.x, g.p1.y, sr, L.t_limit, lh);
						lm = g.material;
					} else {
						make_hit<F>(S, L.skip, sr, L.t_limit, lh, lm);
					}
				} else {
					make_hit<F>(S, L.skip, sr, L.t_limit, lh, lm);
					g = load_prim<F>(S, L.skip);
				}
				const float p = prim_scattering_pdf<F>(g, hit.point, L.l_wi, lh);
				if (p > 0.0f) {
					le = mat_get_emission<F>(S, lm, lh, L.l_wi);
					l_pdf = p * L.pdf_multiplier;
					valid = true;
				}
			}
			if (valid) { // mis.rs:39-43
				const float m_pdf = mat_scattering_pdf<F>(S, mat, hit, wo, L.l_wi, known_path);
				const float mis_weight = power_heuristic(l_pdf, m_pdf);
				outp = outp + thr * mat_eval<F>(S, mat, hit, wo, L.l_wi, known_path, PS) * mis_weight * le / l_pdf;
			}
		}
		RT_SECTION(10); // SCATTER: the light's contribution
		// ---- material sampling  mis.rs:46-49.  scatter_ray reads only the incoming direction of
		// the ray that produced `hit`, which is `wo` (mis.rs:21,82); `ray` held the shadow ray. ----
		ray.d = wo;
		if (mat_scatter_ray<F>(S, mat, ray, hit, rng, known_path))
			finalize(true);
		else
			begin_walk(false);
	
	};

	// NARROW -- fine schedule, irregular ray: the whole walk over the two-child tree, here and now (rare: a
	// direction with an exactly zero component).  trace_closest / trace_any pick that tree for such rays themselves.
	auto do_narrow = [&]() {
		if (any_hit) {
			best_prim = trace_any<F, PRUNE, true>(S, S_global, SM, ray, stk, PL.t_limit, PL.skip) ? 0u : kNoPrim;
			ph = PH_SCATTER;
		} else {
			trace_closest<F, PRUNE, true>(S, S_global, SM, ray, stk, best_t, best_prim);
			ph = PH_SHADE;
		}
		node = kRefDone;
		sp = 0;
	};

	// coarse schedule: run the pending walk of this lane to its end with the tight while-while loops
	// of rt_intersect.h (closest walk in the TRACE super-phase, shadow walk in the LIGHT super-phase)
	auto walk_closest_pending = [&]() {
		if (ph == PH_NODE && !any_hit) {
			trace_closest<F, PRUNE>(S, S_global, SM, ray, stk, best_t, best_prim, PS);
			ph = PH_SHADE;
		}
	};
	auto walk_shadow_pending = [&](const LightCtx &L, const Ray &sr) {
		if (ph == PH_NODE && any_hit) {
			best_prim = trace_any<F, PRUNE>(S, S_global, SM, sr, stk, L.t_limit, L.skip, PS) ? 0u : kNoPrim;
			ph = PH_SCATTER;
		}
	};

	// ---- XCHG, fine schedule: the lane's whole state as one record, and the trade itself (see kXchgRecWords above) ----
	auto record_store = [&](uint32_t *rec) {
		uint4 *d = reinterpret_cast<uint4 *>(rec);
		const uint32_t fl = (hit.has_uv ? 1u : 0u) | (hit.out ? 2u : 0u) | (primary ? 4u : 0u) | (any_hit ? 8u : 0u) | (PL.have_shadow ? 16u : 0u) |
		                    (PL.shadow_is_sky ? 32u : 0u);
		d[0] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
		d[1] = make_uint4(__float_as_uint(thr.x), __float_as_uint(thr.y), __float_as_uint(thr.z), __float_as_uint(outp.x));
		d[2] = make_uint4(__float_as_uint(outp.y), __float_as_uint(outp.z), __float_as_uint(wo.x), __float_as_uint(wo.y));
		d[3] = make_uint4(__float_as_uint(wo.z), __float_as_uint(hit.t), __float_as_uint(hit.point.x), __float_as_uint(hit.point.y));
		d[4] = make_uint4(__float_as_uint(hit.point.z), __float_as_uint(hit.error.x), __float_as_uint(hit.error.y), __float_as_uint(hit.error.z));
		d[5] = make_uint4(__float_as_uint(hit.normal.x), __float_as_uint(hit.normal.y), __float_as_uint(hit.normal.z), __float_as_uint(hit.uvx));
		d[6] = make_uint4(__float_as_uint(hit.uvy), fl, mat, depth);
		d[7] = make_uint4(ray_count, __float_as_uint(mean.x), __float_as_uint(mean.y), __float_as_uint(mean.z));
		d[8] = make_uint4(sample_local, out_index, pixel_index, px);
		d[9] = make_uint4(py, chunk_begin, chunk_n, (uint32_t)ph);
		d[10] = make_uint4(__float_as_uint(ray.o.x), __float_as_uint(ray.o.y), __float_as_uint(ray.o.z), __float_as_uint(ray.d.x));
		d[11] = make_uint4(__float_as_uint(ray.d.y), __float_as_uint(ray.d.z), __float_as_uint(ray.inv.x), __float_as_uint(ray.inv.y));
		d[12] = make_uint4(__float_as_uint(ray.inv.z), __float_as_uint(ray.shear.x), __float_as_uint(ray.shear.y), __float_as_uint(ray.shear.z));
		d[13] = make_uint4(__float_as_uint(best_t), best_prim, __float_as_uint(PL.t_limit), PL.skip);
		d[14] = make_uint4(__float_as_uint(PL.l_wi.x), __float_as_uint(PL.l_wi.y), __float_as_uint(PL.l_wi.z), __float_as_uint(PL.pdf_multiplier));
	};
	auto record_load = [&](const uint32_t *rec) {
		const uint4 *d = reinterpret_cast<const uint4 *>(rec);
		const uint4 a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3], a4 = d[4], a5 = d[5], a6 = d[6], a7 = d[7], a8 = d[8], a9 = d[9];
		const uint4 a10 = d[10], a11 = d[11], a12 = d[12], a13 = d[13], a14 = d[14];
		rng.s0 = a0.x; rng.s1 = a0.y; rng.s2 = a0.z; rng.s3 = a0.w;
		thr = v3(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z));
		outp = v3(__uint_as_float(a1.w), __uint_as_float(a2.x), __uint_as_float(a2.y));
		wo = v3(__uint_as_float(a2.z), __uint_as_float(a2.w), __uint_as_float(a3.x));
		hit.t = __uint_as_float(a3.y);
		hit.point = v3(__uint_as_float(a3.z), __uint_as_float(a3.w), __uint_as_float(a4.x));
		hit.error = v3(__uint_as_float(a4.y), __uint_as_float(a4.z), __uint_as_float(a4.w));
		hit.normal = v3(__uint_as_float(a5.x), __uint_as_float(a5.y), __uint_as_float(a5.z));
		hit.err_dot = dot(vabs(hit.normal), hit.error); // (not part of the record: re-formed from what is)
		hit.uvx = __uint_as_float(a5.w);
		hit.uvy = __uint_as_float(a6.x);
		hit.has_uv = (a6.y & 1u) != 0u; hit.out = (a6.y & 2u) != 0u; primary = (a6.y & 4u) != 0u;
		any_hit = (a6.y & 8u) != 0u; PL.have_shadow = (a6.y & 16u) != 0u; PL.shadow_is_sky = (a6.y & 32u) != 0u;
		mat = a6.z; depth = a6.w;
		ray_count = a7.x;
		mean = v3(__uint_as_float(a7.y), __uint_as_float(a7.z), __uint_as_float(a7.w));
		sample_local = a8.x; out_index = a8.y; pixel_index = a8.z; px = a8.w;
		py = a9.x; chunk_begin = a9.y; chunk_n = a9.z; ph = (int)a9.w;
		ray.o = v3(__uint_as_float(a10.x), __uint_as_float(a10.y), __uint_as_float(a10.z));
		ray.d = v3(__uint_as_float(a10.w), __uint_as_float(a11.x), __uint_as_float(a11.y));
		ray.inv = v3(__uint_as_float(a11.z), __uint_as_float(a11.w), __uint_as_float(a12.x));
		ray.shear = v3(__uint_as_float(a12.y), __uint_as_float(a12.z), __uint_as_float(a12.w));
		best_t = __uint_as_float(a13.x); best_prim = a13.y; PL.t_limit = __uint_as_float(a13.z); PL.skip = a13.w;
		PL.l_wi = v3(__uint_as_float(a14.x), __uint_as_float(a14.y), __uint_as_float(a14.z));
		PL.pdf_multiplier = __uint_as_float(a14.w);
		node = kRefDone; // (a record never holds a walk in progress)
		sp = 0;
	};
	const bool xchg_shader = wave < kXchgShaderWaves;
	auto xchg_try_lock = [&]() -> bool {
		uint32_t got = 0u;
		if (lane == 0u) {
			uint32_t expect = 0u;
			got = __hip_atomic_compare_exchange_strong(&pool[0], &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
		}
		if (__shfl(got, 0) == 0u)
			return false;
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		return true;
	};
	auto xchg_unlock = [&]() {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0u)
			__hip_atomic_store(&pool[0], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	auto exchange_fine = [&]() {
		const uint32_t slots = kXchgFineSlots;
		// the shader gives walks to start (W) and takes walk results (R); a walker the other way round
		const uint32_t iG = xchg_shader ? 1u : 2u, iT = xchg_shader ? 2u : 1u, iOther = xchg_shader ? 4u : 3u;
		uint32_t *stG = pool + 8 + (xchg_shader ? 0u : 64u), *stT = pool + 8 + (xchg_shader ? 64u : 0u);
		uint32_t *xmap = pool + kXchgFineHdr + wave * 64u; // this wave's scratch: slot of the q-th claim
		uint32_t *recW = pool + kXchgFineHdr + (blockDim.x >> 6) * 64u, *recR = recW + slots * kXchgRecWords;
		uint32_t *recG = xchg_shader ? recW : recR, *recT = xchg_shader ? recR : recW;
		const unsigned long long mGive = xchg_shader ? __ballot(ph == PH_WALK) : __ballot(ph == PH_SHADE || ph == PH_SCATTER);
		const unsigned long long mFree = __ballot(ph == PH_NEED_PIXEL || ph == PH_DONE);
		const uint32_t n_give_want = (uint32_t)__popcll(mGive), n_free = (uint32_t)__popcll(mFree);
		const uint32_t peekG = __hip_atomic_load(&pool[iG], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		const uint32_t peekT = __hip_atomic_load(&pool[iT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		const uint32_t other = __hip_atomic_load(&pool[iOther], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		// A trade is a SWAP, one record given for one taken: lanes vacated by a gift without a return would fetch fresh
		// pixels, and the extra states would fill both pools until nothing moves.  Gifts without a return only prime the
		// pools (while the two together hold less than one pool's worth); lanes with nothing take without giving.
		const bool may_swap = other != 0u && n_give_want >= kXchgMinLanes && peekG < slots && peekT != 0u;
		const bool may_prime = other != 0u && n_give_want >= kXchgMinLanes && peekG + peekT + kXchgMinLanes <= slots;
		const bool may_take = peekT != 0u && n_free != 0u;
		if (!may_swap && !may_prime && !may_take)
			return;
		if (!xchg_try_lock())
			return;
		// ---- under the lock: claim slots (state words only) ----
		const uint32_t sG = lane < slots ? stG[lane] : (uint32_t)XS_BUSY, sT = lane < slots ? stT[lane] : (uint32_t)XS_BUSY;
		const unsigned long long m_empty = __ballot(sG == XS_EMPTY), m_full = __ballot(sT == XS_FULL);
		const uint32_t other_now = pool[iOther], parked = pool[1] + pool[2];
		const uint32_t n_can_give = (other_now != 0u && n_give_want >= kXchgMinLanes) ? min(n_give_want, (uint32_t)__popcll(m_empty)) : 0u;
		const uint32_t n_swap = min(n_can_give, (uint32_t)__popcll(m_full));
		const uint32_t n_prime = min(n_can_give - n_swap, parked < slots ? slots - parked : 0u);
		const uint32_t n_give = n_swap + n_prime;
		const uint32_t n_take = n_swap + min((uint32_t)__popcll(m_full) - n_swap, n_free);
		const unsigned long long below = (1ull << lane) - 1ull;
		const uint32_t qE = (uint32_t)__popcll(m_empty & below), qF = (uint32_t)__popcll(m_full & below);
		const bool claimE = (m_empty >> lane & 1ull) != 0ull && qE < n_give, claimF = (m_full >> lane & 1ull) != 0ull && qF < n_take;
		if (claimE)
			stG[lane] = XS_BUSY;
		if (claimF)
			stT[lane] = XS_BUSY;
		if (lane == 0u) {
			pool[iG] = pool[iG] + n_give;
			pool[iT] = pool[iT] - n_take;
		}
		xchg_unlock();
		// ---- outside it: move the records ----
		const uint32_t rG = (uint32_t)__popcll(mGive & below);
		const bool giver = (mGive >> lane & 1ull) != 0ull && rG < n_give;
		if (claimE)
			xmap[qE] = lane;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		uint32_t slotG = 0u;
		if (giver) {
			slotG = xmap[rG];
			record_store(recG + slotG * kXchgRecWords);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (giver)
			__hip_atomic_store(&stG[slotG], (uint32_t)XS_FULL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		__builtin_amdgcn_wave_barrier();
		if (claimF)
			xmap[qF] = lane;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// who takes: the lanes that just gave their state away first, then the lanes that had nothing
		const bool is_free = (mFree >> lane & 1ull) != 0ull;
		const uint32_t rT = (giver && rG < n_swap) ? rG : (is_free ? n_swap + (uint32_t)__popcll(mFree & below) : 0xFFFFFFFFu);
		const bool taker = rT < n_take;
		uint32_t slotT = 0u;
		if (taker) {
			slotT = xmap[rT];
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			record_load(recT + slotT * kXchgRecWords);
		} else if (giver) {
			ph = PH_NEED_PIXEL; // the lane is vacant (its record primed the pool)
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (taker)
			__hip_atomic_store(&stT[slotT], (uint32_t)XS_EMPTY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
	};
	// the wave has nothing left: it may go once the pool it takes from is empty -- and says so under the lock, so that
	// the other role stops giving to it (nothing can be parked for a wave that has left)
	auto xchg_try_exit = [&]() -> bool {
		if (!xchg_try_lock())
			return false;
		const uint32_t *stT = pool + 8 + (xchg_shader ? 64u : 0u);
		const bool empty = __ballot(lane < kXchgFineSlots && stT[lane] != XS_EMPTY) == 0ull;
		if (empty && lane == 0u)
			pool[xchg_shader ? 3u : 4u] -= 1u;
		xchg_unlock();
		return empty;
	};

	// XCHG -- trade with the workgroup's pool before the wave votes.  Between iterations of the coarse MIS schedule a
	// lane is in exactly one of four states: PH_GEN (a pixel between samples), PH_LIGHT (a path between bounces),
	// PH_NEED_PIXEL, PH_DONE.  The wave picks the super-phase it could run FULLER after a trade -- BOUNCE by swapping its
	// pixels for parked paths, PRIMARY by parking its paths and taking parked (or fresh) pixels -- so the level of the
	// pool steers the waves of a workgroup into the two roles in the proportion the scene asks for.
	auto exchange = [&]() {
		const uint32_t slots = P.xchg_slots;
		const unsigned long long mB = __ballot(ph == PH_LIGHT), mP = __ballot(ph == PH_GEN);
		const unsigned long long mD = __ballot(ph == PH_DONE), mE = mD | __ballot(ph == PH_NEED_PIXEL);
		const uint32_t nb = (uint32_t)__popcll(mB), np = (uint32_t)__popcll(mP), ne = (uint32_t)__popcll(mE);
		const uint32_t nn = ne - (uint32_t)__popcll(mD);
		const bool fresh = mD == 0ull; // the frame still has unclaimed pixels (as far as this wave knows)
		const uint32_t peekB = __hip_atomic_load(&pool[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		const uint32_t peekP = __hip_atomic_load(&pool[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		// lanes each super-phase would run on after the trade
		const uint32_t bounce_lanes = nb + min(min(np, slots - peekP) + ne, peekB);
		const uint32_t primary_lanes = np + nn + min(nb, fresh ? slots - peekB : min(slots - peekB, peekP));
		const bool will_bounce = (np + nn) == 0u || (bounce_lanes >= kLightPhaseThreshold && bounce_lanes >= primary_lanes);
		bool want;
		if (will_bounce)
			want = ((np >= kXchgMinLanes || ne != 0u) && peekB != 0u) || (nb == 0u && np == 0u && peekP != 0u);
		else
			want = (nb >= kXchgMinLanes && peekB < slots && (fresh || peekP != 0u)) || (ne != 0u && peekP != 0u);
		if (!want)
			return;
		uint32_t got = 0u;
		if (lane == 0u) {
			uint32_t expect = 0u;
			got = __hip_atomic_compare_exchange_strong(&pool[0], &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
		}
		if (__shfl(got, 0) == 0u)
			return; // somebody else is trading: vote with what the wave has
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		uint32_t cB = pool[1], cP = pool[2];
		uint32_t *poolB = pool + 4, *poolP = pool + 4 + kXchgBStride * slots;
		auto park_path = [&](uint32_t slot) {
			uint4 *d = reinterpret_cast<uint4 *>(poolB + slot * kXchgBStride);
			const uint32_t fl = (hit.has_uv ? 1u : 0u) | (hit.out ? 2u : 0u) | (primary ? 4u : 0u);
			d[0] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
			d[1] = make_uint4(__float_as_uint(thr.x), __float_as_uint(thr.y), __float_as_uint(thr.z), __float_as_uint(outp.x));
			d[2] = make_uint4(__float_as_uint(outp.y), __float_as_uint(outp.z), __float_as_uint(wo.x), __float_as_uint(wo.y));
			d[3] = make_uint4(__float_as_uint(wo.z), __float_as_uint(hit.t), __float_as_uint(hit.point.x), __float_as_uint(hit.point.y));
			d[4] = make_uint4(__float_as_uint(hit.point.z), __float_as_uint(hit.error.x), __float_as_uint(hit.error.y), __float_as_uint(hit.error.z));
			d[5] = make_uint4(__float_as_uint(hit.normal.x), __float_as_uint(hit.normal.y), __float_as_uint(hit.normal.z), __float_as_uint(hit.uvx));
			d[6] = make_uint4(__float_as_uint(hit.uvy), fl, mat, depth);
			d[7] = make_uint4(ray_count, __float_as_uint(mean.x), __float_as_uint(mean.y), __float_as_uint(mean.z));
			d[8] = make_uint4(sample_local, out_index, pixel_index, px);
			d[9] = make_uint4(py, chunk_begin, chunk_n, 0u);
		};
		auto take_path = [&](uint32_t slot) {
			const uint4 *d = reinterpret_cast<const uint4 *>(poolB + slot * kXchgBStride);
			const uint4 a0 = d[0], a1 = d[1], a2 = d[2], a3 = d[3], a4 = d[4], a5 = d[5], a6 = d[6], a7 = d[7], a8 = d[8], a9 = d[9];
			rng.s0 = a0.x; rng.s1 = a0.y; rng.s2 = a0.z; rng.s3 = a0.w;
			thr = v3(__uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z));
			outp = v3(__uint_as_float(a1.w), __uint_as_float(a2.x), __uint_as_float(a2.y));
			wo = v3(__uint_as_float(a2.z), __uint_as_float(a2.w), __uint_as_float(a3.x));
			hit.t = __uint_as_float(a3.y);
			hit.point = v3(__uint_as_float(a3.z), __uint_as_float(a3.w), __uint_as_float(a4.x));
			hit.error = v3(__uint_as_float(a4.y), __uint_as_float(a4.z), __uint_as_float(a4.w));
			hit.normal = v3(__uint_as_float(a5.x), __uint_as_float(a5.y), __uint_as_float(a5.z));
		hit.err_dot = dot(vabs(hit.normal), hit.error); // (not part of the record: re-formed from what is)
			hit.uvx = __uint_as_float(a5.w);
			hit.uvy = __uint_as_float(a6.x);
			hit.has_uv = (a6.y & 1u) != 0u; hit.out = (a6.y & 2u) != 0u; primary = (a6.y & 4u) != 0u;
			mat = a6.z; depth = a6.w;
			ray_count = a7.x;
			mean = v3(__uint_as_float(a7.y), __uint_as_float(a7.z), __uint_as_float(a7.w));
			sample_local = a8.x; out_index = a8.y; pixel_index = a8.z; px = a8.w;
			py = a9.x; chunk_begin = a9.y; chunk_n = a9.z;
			ph = PH_LIGHT;
		};
		auto park_pixel = [&](uint32_t slot) {
			uint4 *d = reinterpret_cast<uint4 *>(poolP + slot * kXchgPStride);
			d[0] = make_uint4(__float_as_uint(mean.x), __float_as_uint(mean.y), __float_as_uint(mean.z), sample_local);
			d[1] = make_uint4(out_index, pixel_index, px, py);
			d[2] = make_uint4(chunk_begin, chunk_n, 0u, 0u);
		};
		auto take_pixel = [&](uint32_t slot) {
			const uint4 *d = reinterpret_cast<const uint4 *>(poolP + slot * kXchgPStride);
			const uint4 a0 = d[0], a1 = d[1], a2 = d[2];
			mean = v3(__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z));
			sample_local = a0.w; out_index = a1.x; pixel_index = a1.y; px = a1.z; py = a1.w;
			chunk_begin = a2.x; chunk_n = a2.y;
			ph = PH_GEN;
		};
		const unsigned long long below = (1ull << lane) - 1ull;
		const uint32_t rB = (uint32_t)__popcll(mB & below), rP = (uint32_t)__popcll(mP & below), rE = (uint32_t)__popcll(mE & below);
		const bool is_e = (mE >> lane & 1ull) != 0ull;
		if (!will_bounce) {
			// PRIMARY next: the wave's paths between bounces would idle.  Park them; a vacated lane takes a parked pixel if
			// there is one, else asks for a fresh one -- but only while fresh pixels exist: after that a path is parked
			// only against a pixel taken, so the pool cannot fill with work nobody is left to run.
			uint32_t can = nb >= kXchgMinLanes ? min(nb, slots - cB) : 0u;
			if (!fresh)
				can = min(can, cP);
			if (ph == PH_LIGHT && rB < can) {
				park_path(cB + rB);
				if (rB < cP)
					take_pixel(cP - 1u - rB);
				else
					ph = PH_NEED_PIXEL;
			}
			cB += can;
			cP -= min(can, cP);
			const uint32_t te = min(ne, cP); // lanes with nothing to do take parked pixels
			if (is_e && rE < te)
				take_pixel(cP - 1u - rE);
			cP -= te;
		} else {
			// BOUNCE next: the wave's pixels between samples would idle.  Each swaps with a parked path, one for one; lanes
			// with nothing to do just take paths.
			const uint32_t swaps = np >= kXchgMinLanes ? min(min(np, cB), slots - cP) : 0u;
			if (ph == PH_GEN && rP < swaps) {
				park_pixel(cP + rP);
				take_path(cB - 1u - rP);
			}
			cB -= swaps;
			cP += swaps;
			const uint32_t te = min(ne, cB);
			if (is_e && rE < te)
				take_path(cB - 1u - rE);
			cB -= te;
			if (nb == 0u && np == 0u && te == 0u) { // the tail of the frame: a wave with nothing at all takes pixels too
				const uint32_t tp = min(ne, cP);
				if (is_e && rE < tp)
					take_pixel(cP - 1u - rE);
				cP -= tp;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0u) {
			pool[1] = cB;
			pool[2] = cP;
			__hip_atomic_store(&pool[0], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	};

	uint32_t wq_next = 0, wq_end = 0; // wave-uniform: this wave's private range of work items
	// ... and, when a claim lies inside one tile (DevRenderParams::tile_log2_w), that tile's origin (x | y << 16) and the index,
	// in this shard's pixel order, of the first pixel the claim covers: worked out once per claim, not once per item
	uint32_t wq_xy = 0, wq_pbase = 0;
	// Work acquisition when a claim lies inside a tile.  Two things are wrong with decoding every item on its own in chunk-major
	// order (item = chunk w / n_work of pixel w % n_work, the general path below) once sample_split > 1:
	//  * the decode -- which tile, which chunk, which passes -- is four 32-bit and two 64-bit integer divisions, some 600
	//    instructions the whole wave executes whenever ANY of its lanes wants an item;
	//  * chunk-major order sweeps the image once per chunk, and at the end of every sweep a wave's last lanes are still on the
	//    expensive pixels at the bottom of the image while the others start on the cheap ones at the top: fewer than
	//    kLightPhaseThreshold lanes wait for BOUNCE, the others regenerate for ever, and the waiting lanes are stranded until the
	//    wave reaches expensive pixels again (config 2, stats build: PRIMARY runs with 30 lanes instead of 51; 91 -> 101 ms
	//    whatever the split).
	// Here a claim of 64 items is 64 / S consecutive pixels of ONE tile times their S chunks (S a power of two <= 64), the
	// chunks of a pixel in neighbouring lanes: the image is swept once, like S = 1, the divisions are done once per claim on
	// wave-uniform values, and a lane's own part is masks and shifts.  What a pixel's chunks return does not depend on who folds
	// them when, so the frame is the same as under the general order.
	auto acquire_coarse = [&]() {
		const unsigned long long need = __ballot(ph == PH_NEED_PIXEL);
		// Serving an acquire event costs the WHOLE wave a round of scalar loads and some sixty instructions, and with short items
		// (sample_split) an event is due in every third or fourth iteration, half of them for a single lane.  So a few needy lanes
		// wait for company -- an idle lane costs 1/64 of an iteration -- as long as the wave has other work to vote on.
		if (kAcquireBatch > 1u && (uint32_t)__popcll(need) < kAcquireBatch && __ballot(ph == PH_GEN || ph == PH_LIGHT) != 0ull)
			return;
		const KArgs k = kargs();
		const bool tiled = P.tile_log2_w != 0xFFFFFFFFu; // (wave-uniform) the tiled order above, or the general one (any tile size, any split)
		// The wave-uniform part (scalar registers only) sits behind the wave-uniform tests; the lanes' part below does NOT: it is ONE
		// divergent region, shared by the two orders, that no lane may enter.  Behind `if (need == 0) return;`, and with a region per
		// order, the lane state of the needy lanes (pixel, chunk, running sum, phase: 15 registers) had paths round the regions on which
		// it was untouched, and the compiler kept it in two register sets: 15 v_mov at the top and 15 at the bottom of EVERY iteration
		// of the persistent loop (round 4).
		uint32_t avail = 0, old_next = 0, old_xy = 0, old_pbase = 0, base = 0, log2_w = 0, log2_s = 0;
		if (need != 0ull) {
			const uint32_t n = (uint32_t)__popcll(need);
#if defined(RT_STATS) && !defined(RT_STATS_NO_HIST)
			if (lane == 0u)
				atomicAdd(&g_hist[2][n], 1ull);
#endif
			avail = wq_end - wq_next;
			old_next = wq_next, old_xy = wq_xy, old_pbase = wq_pbase;
			if (tiled)
				log2_w = k->P.tile_log2_w & 7u, log2_s = k->P.tile_log2_w >> 8;
			base = wq_end;
			if (avail < n) { // the range runs short: claim the next 64 items
				const int leader = __ffsll((long long)need) - 1;
				uint32_t claimed = 0;
				if ((int)lane == leader)
					claimed = atomicAdd(k->work_counter, kClaim);
				// (readlane, not __shfl: a shuffle's result counts as divergent, and through `base` the wave's range would live in
				// vector registers for the whole loop)
				base = (uint32_t)__builtin_amdgcn_readlane((int)claimed, leader);
				if (tiled) {
					const uint32_t b = base >> 6;              // claim number
					const uint32_t kt = b >> log2_s;           // ... lies in this shard's kt-th tile
					const uint32_t sub = b & ((1u << log2_s) - 1u); // ... and is that tile's sub-th group of 64 / S pixels
					const uint32_t tile = k->P.shard_index + kt * k->P.shard_count;
					const uint32_t ty = tile / k->P.tiles_x;
					const uint32_t tx = tile - ty * k->P.tiles_x;
					wq_xy = (tx << log2_w) | ((ty * k->P.tile_h) << 16);
					wq_pbase = (kt << 6) + (sub << (6u - log2_s));
				}
				// the leftovers go first, the rest comes from the new claim
				wq_next = base + (n - avail);
				wq_end = base + kClaim;
			} else {
				wq_next += n;
			}
		}
		if (ph == PH_NEED_PIXEL) {
			const uint32_t r = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
			const bool from_old = r < avail;
			const uint32_t w = from_old ? old_next + r : base + (r - avail);
			const uint32_t spp = k->P.spp;
			bool inside;
			if (tiled) {
				const uint32_t xy = from_old ? old_xy : wq_xy;
				const uint32_t in = w & 63u;
				// (the other single-sweep layout -- a claim = all 64 pixels of the tile for ONE chunk, claims tile-major -- measured the
				// same to 0.3 % at every split: profiles/r03_ab_logs/r05v_split_ab.log)
				const uint32_t wp = (from_old ? old_pbase : wq_pbase) + (in >> log2_s); // the pixel, in this shard's order
				const uint32_t c = in & ((1u << log2_s) - 1u);                            // ... and which of its chunks
				const uint32_t in_tile = wp & 63u;
				px = (xy & 0xFFFFu) + (in_tile & ((1u << log2_w) - 1u));
				py = (xy >> 16) + (in_tile >> log2_w);
				inside = px < k->P.width && py < k->P.height;
				pixel_index = py * k->P.width + px;
				chunk_begin = (uint32_t)(((uint64_t)c * spp) >> log2_s); // = c * spp / S (rt_hip.h sample_split)
				chunk_n = (uint32_t)(((uint64_t)(c + 1u) * spp) >> log2_s) - chunk_begin;
				// chunk sums go to the partial buffer chunk-major (combine_chunks_kernel); whole pixels to the frame or the packed shard
				out_index = log2_s != 0u ? c * k->P.n_work + wp : (k->P.shard_layout ? wp : pixel_index);
			} else {
				const uint32_t split = k->P.sample_split, n_work = k->P.n_work;
				inside = work_to_pixel(P, split > 1u ? w % n_work : w, px, py);
				pixel_index = py * k->P.width + px;
				if (split > 1u) {
					// sample_split (rt_hip.h): this item is chunk c of its pixel; its sum goes to the
					// partial buffer (chunk-major) and combine_chunks_kernel folds the chunks in order
					const uint32_t c = w / n_work;
					chunk_begin = (uint32_t)(((uint64_t)c * spp) / split);
					chunk_n = (uint32_t)(((uint64_t)(c + 1u) * spp) / split) - chunk_begin;
					out_index = w;
				} else {
					chunk_begin = 0u;
					chunk_n = spp;
					out_index = k->P.shard_layout ? w : pixel_index;
				}
			}
			if (w >= k->P.n_items) {
				ph = PH_DONE;
			} else if (inside) {
				sample_local = 0;
				mean = v3s(0.0f);
				ph = PH_GEN;
			} // else: padding of an edge tile; ask again next iteration
		}
	};
	auto acquire = [&]() {
	if constexpr (!FINE) { // (the fine kernels keep the general order below: 1.5 % slower with the tiled one on 1 M triangles, r05p)
		acquire_coarse();
		return;
	}
	// ---- work acquisition.  Lanes that ran out of samples are served from a wave-private range of
	// work items [wq_next, wq_end); when that runs short the wave claims kClaim more items with ONE
	// atomic on the global counter (a single word sustains only ~88 dequeues/us on this chip, and
	// with sample_split there can be tens of millions of items).  All of this is wave-uniform. ----
	{
		const unsigned long long need = __ballot(ph == PH_NEED_PIXEL);
		if (need != 0ull) {
			const KArgs k = kargs(); // the tile geometry is needed once per pixel: read it here, not in the prologue
			const DevRenderParams P = k->P;
			const uint32_t n = (uint32_t)__popcll(need);
			const uint32_t avail = wq_end - wq_next;
			uint32_t base = wq_end;
			if (avail < n) {
				const int leader = __ffsll((long long)need) - 1;
				uint32_t claimed = 0;
				if ((int)lane == leader)
					claimed = atomicAdd(k->work_counter, kClaim);
				// (fine schedule only since round 4.  A shuffle's result counts as divergent, so the wave's range [wq_next, wq_end) lives
				// in vector registers here; with readlane, as in acquire_coarse, it moves to scalar ones -- and the two mesh workloads
				// get 0.8 % / 0.7 % SLOWER, same box, profiles/r04v_mesh_ab.log: the fine kernels are short of scalar registers, not vector ones)
				base = __shfl(claimed, leader);
			}
			if (ph == PH_NEED_PIXEL) {
				const uint32_t r = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
				const uint32_t w = r < avail ? wq_next + r : base + (r - avail);
				if (w >= P.n_items) {
					ph = PH_DONE;
				} else if (work_to_pixel(P, P.sample_split > 1u ? w % P.n_work : w, px, py)) {
					pixel_index = py * P.width + px;
					if (P.sample_split > 1u) {
						// sample_split (rt_hip.h): this item is chunk c of its pixel; its mean goes to the
						// partial buffer (chunk-major) and combine_chunks_kernel folds the chunks in order
						const uint32_t c = w / P.n_work;
						chunk_begin = (uint32_t)(((uint64_t)c * P.spp) / P.sample_split);
						chunk_n = (uint32_t)(((uint64_t)(c + 1u) * P.spp) / P.sample_split) - chunk_begin;
						out_index = w;
					} else {
						out_index = P.shard_layout ? w : pixel_index;
					}
					sample_local = 0;
					mean = v3s(0.0f);
					ph = PH_GEN;
				} // else: padding of an edge tile; ask again next iteration
			}
			if (avail < n) { // the leftovers went first, the rest came from the new block
				wq_next = base + (n - avail);
				wq_end = base + kClaim;
			} else {
				wq_next += n;
			}
		}
	}
	};
	if constexpr (FINE)
	for (;;) {
		if (XCHG) {
			exchange_fine();
			// walks start where they are now -- except in the shading wave, which holds the ones it could not give away
			// until the walkers have made room (they do not depend on it for that), or starts them itself once no walker is left
			if (ph == PH_WALK && (!xchg_shader || __hip_atomic_load(&pool[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u))
				start_walk();
		}
		acquire();
		if (FINE) {
			// ---- big trees: every step is its own phase; run the one most lanes wait for
			// (ties: the later pipeline stage) ----
			// NODE steps are the bulk of the work (a walk is tens to hundreds of them, 8 per vote), every other
			// phase is short.  So the short phases are DRAINED eagerly -- any of them runs as soon as
			// kDrainLanes lanes wait for it, later pipeline stages first -- which keeps lanes flowing back
			// into NODE and NODE's occupancy high; NODE runs otherwise; when no NODE lane is left the fullest
			// remaining phase runs.  (Pure arg-max left NODE at 32/64 lanes on 1 M triangles.)
			uint32_t cnt[PH_COUNT];
#pragma unroll
			for (int k = 0; k < PH_COUNT; ++k)
				cnt[k] = (uint32_t)__popcll(__ballot(ph == k));
			int run = -1;
			uint32_t best_n = 0;
#pragma unroll
			for (int k = PH_COUNT - 1; k >= 0; --k) {
				if (XCHG && xchg_shader)
					break; // the shading wave has nothing to keep flowing: it runs its fullest phase (below)
				if (k != PH_NODE && run < 0 && cnt[k] >= (k == PH_NARROW ? 1u : (k == PH_LEAF ? kDrainLanes : kDrainLanesHeavy))) {
					run = k;
					best_n = cnt[k];
				}
			}
			if (run < 0 && cnt[PH_NODE] > 0u && !(XCHG && xchg_shader)) {
				run = PH_NODE;
				best_n = cnt[PH_NODE];
			}
			if (run < 0) {
#pragma unroll
				for (int k = 0; k < PH_COUNT; ++k) {
					if (cnt[k] >= best_n && cnt[k] > 0u) {
						best_n = cnt[k];
						run = k;
					}
				}
			}
			if (run < 0) {
				if (XCHG && __ballot(ph == PH_WALK) != 0ull) { // (the shading wave, holding walks for the walkers)
					__builtin_amdgcn_s_sleep(4);
					continue;
				}
				if (__ballot(ph == PH_NEED_PIXEL) == 0ull) { // every lane is PH_DONE
					if (XCHG && !xchg_try_exit()) {
						__builtin_amdgcn_s_sleep(4);
						continue; // records are still parked for this wave (or the lock was taken): look again
					}
					break;
				}
				continue; // only edge-tile padding was handed out: ask again
			}
#ifdef RT_STATS
			if (lane == 0u) {
				st_fine_iters[run] += 1;
				st_fine_active[run] += best_n;
			}
#endif
#ifdef RT_STATS
			if (lane == 0u) { // wall-clock share per fine phase: slot 6 = vote + work acquisition, 0..5 = the phase run last
				const unsigned long long now_ = wall_clock64();
				st_fine_clock[PH_COUNT] += now_ - st_mark;
				st_mark = now_;
			}
#endif
			// (The arms stay an else-if chain.  The chain's join costs about 135 v_mov per iteration -- every arm leaves the lane state
			// where it suited that arm -- and written as a SEQUENCE of guarded regions, the form that freed the coarse loop of its
			// copies, the fine MIS kernel has 715 instead of 1010 v_mov but spills 71 VGPRs: 1 M triangles 947 -> 1 183 ms, 10 M
			// 992 -> 1 455 ms, same box, profiles/r04x_fine_sequence_ab.log)
			if (run == PH_NODE) {
				// a few node steps per vote: a walk is tens to hundreds of them and the vote is not free
#pragma unroll 1
				for (int step = 0; step < kNodeStepsPerVote; ++step)
					if (ph == PH_NODE)
						do_node();
			} else if (run == PH_LEAF) {
				if (ph == PH_LEAF)
					do_leaf();
			} else if (run == PH_GEN) {
				if (ph == PH_GEN)
					do_gen();
			} else if (run == PH_SHADE) {
				if (ph == PH_SHADE)
					do_shade(std::integral_constant<int, 0>{});
			} else if (run == PH_LIGHT) {
				if (ph == PH_LIGHT)
					do_light(PL, ray);
			} else if (run == PH_NARROW) {
				if (ph == PH_NARROW)
					do_narrow();
			} else {
				if (ph == PH_SCATTER)
					do_scatter(PL, ray);
			}
#ifdef RT_STATS
			if (lane == 0u) {
				const unsigned long long now_ = wall_clock64();
				st_fine_clock[run] += now_ - st_mark;
				st_mark = now_;
			}
#endif
		}
	}

	// ---- tiny trees (a walk is a handful of steps): the coarse schedule, a loop of its own with ONE exit and ONE back edge.
	// (As one `for (;;)` shared with the fine schedule, leaving through `break` / `continue` in nested branches, the
	// structurised loop copied the whole loop-carried lane state -- about 45 registers -- twice per iteration.) ----
	if constexpr (!FINE) {
		bool alive = true; // wave-uniform
		// lane-iterations that paths have spent waiting for BOUNCE since it last ran (wave-uniform; see kStarveLimit)
		uint32_t waited = 0;
		while (alive) {
			if (XCHG && METHOD == 1)
				exchange();
			acquire();
			// ---- tiny trees (a walk is a handful of steps): two super-phases.
			//   TRACE = GEN + closest walk + SHADE     LIGHT = LIGHT + shadow walk + SCATTER
			// About half of all SHADE outcomes end the sample; those lanes regenerate and stay in
			// TRACE while continuing paths pile up for LIGHT, which runs once enough lanes wait. ----
			const uint32_t n_light = (uint32_t)__popcll(__ballot(ph == PH_LIGHT));
			// (between two iterations of the MIS loop with its two super-phases a lane is in GEN, LIGHT, NEED_PIXEL or DONE: the walk
			// and both arms of SHADE run inside the super-phase that started them)
			const uint32_t n_trace = (METHOD == 1 && RT_PQ_SPLIT) ? (uint32_t)__popcll(__ballot(ph == PH_GEN))
			                                                      : (uint32_t)__popcll(__ballot(ph == PH_GEN || ph == PH_NODE || ph == PH_LEAF || ph == PH_SHADE));
			// Nothing to run (n_light + n_trace == 0): lanes waiting for a pixel (only edge-tile padding was handed out) ask again;
			// otherwise the wave is done.  kBodyAlways: the body below is NOT skipped in that case -- it finds no lane in any of its
			// phases -- because an `else` arm through which the lane state passes untouched makes the compiler keep that state in
			// two register sets: 25 v_mov at the top of every iteration, 26 - 32 at its bottom, and 101 instead of 76 VGPRs for the
			// config-2 kernel (round 4: 66.3 -> 62.0 ms, same box; profiles/r04q_loop_else_ab.log).  The simple / full variants, which
			// keep one divergent region per step (kFuseRegions below), get SLOWER that way (config 3: 121.3 -> 123.5 ms): they keep the arm.
			constexpr bool kBodyAlways = RT_BODY_ALWAYS(KernelShape<F>::spheres_only);
			if (kBodyAlways) {
				alive = (n_light + n_trace != 0u) || __ballot(ph == PH_NEED_PIXEL) != 0ull;
				if (XCHG && METHOD == 1 && !alive) // parked work left?  (a later push comes from a wave that is still alive and will drain it itself)
					alive = __hip_atomic_load(&pool[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u ||
					        __hip_atomic_load(&pool[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
			}
			if (kBodyAlways || n_light + n_trace != 0u) {
				const bool run_light = (METHOD == 1) && (n_light >= kLightPhaseThreshold || n_trace == 0u || (kStarveLimit != 0u && waited >= kStarveLimit));
				waited = run_light ? 0u : waited + n_light;
	#ifdef RT_STATS
				if (lane == 0u) {
					if (!run_light) {
						st_iters[0] += 1; st_active[0] += n_trace;
						st_hist[((n_trace ? n_trace : 1u) - 1u) >> 3] += 1; // PRIMARY iterations by the lanes that take part, in eighths of a wave
#ifndef RT_STATS_NO_HIST // (three global atomics per iteration: left out when the section clocks are what is wanted)
						atomicAdd(&g_hist[0][n_trace], 1ull);
						atomicAdd(&g_hist[1][n_light], 1ull);
#endif
						st_gen += (unsigned long long)__popcll(__ballot(ph == PH_GEN));
					} else {
						st_iters[1] += 1; st_active[1] += n_light;
#ifndef RT_STATS_NO_HIST
						atomicAdd(&g_hist[3][n_light], 1ull);
#endif
					}
				}
	#endif
				RT_SECTION(0); // vote + work acquisition
				// ONE divergent region per super-phase (a lane that generates a sample also walks and shades it, so the lanes that
				// sit the phase out are merged back once, not after every step: fewer copies of path state) for the spheres-only
				// kernels: config 2 28.46 -> 28.02 ms at 256 spp, same box, three rounds.  The simple / full variants keep one region
				// per step: fused, config 3 measured 42.3 -> 43.3 ms (their spilled registers move into the longer region).
				constexpr bool kFuseRegions = METHOD == 1 && RT_PQ_SPLIT && RT_FUSE_REGIONS(KernelShape<F>::spheres_only);
				bool run_q = run_light;
				if (!run_light) {
					if (kFuseRegions) {
						if (ph == PH_GEN) {
							load_pair_scene();
							do_gen();
							RT_SECTION(1);
							walk_closest_pending();
							RT_SECTION(2);
							do_shade(std::integral_constant<int, 1>{});
						}
						RT_SECTION(3);
						// enough paths wait for BOUNCE now?  Then run it in this iteration: the same sequence of phases the vote at the
						// top of the next iteration would choose, without the trip round the loop in between
						const uint32_t n_light_now = (uint32_t)__popcll(__ballot(ph == PH_LIGHT));
						run_q = n_light_now >= kLightPhaseThreshold;
						if (run_q)
							waited = 0u;
	#ifdef RT_STATS
						if (run_q && lane == 0u) {
							st_iters[1] += 1; st_active[1] += n_light_now;
						}
	#endif
					} else {
						load_pair_scene();
						if (ph == PH_GEN)
							do_gen();
						RT_SECTION(1);
						walk_closest_pending();
						RT_SECTION(2);
						if (ph == PH_SHADE)
							do_shade(std::integral_constant<int, (METHOD == 1 && RT_PQ_SPLIT) ? 1 : 0>{});
						RT_SECTION(3);
					}
				}
				if (run_q) {
					LightCtx L; // loop-local: see LightCtx above
					L.l_wi = v3s(0.0f);
					L.pdf_multiplier = 1.0f;
					L.t_limit = 0.0f;
					L.skip = kNoPrim;
					L.have_shadow = L.shadow_is_sky = false;
					Ray sray;
					sray.o = sray.d = sray.inv = sray.shear = v3s(0.0f);
					if (kFuseRegions) {
						if (ph == PH_LIGHT) {
							load_pair_scene();
							do_light(L, sray);
							RT_SECTION(4);
							walk_shadow_pending(L, sray);
							RT_SECTION(5);
							do_scatter(L, sray);
							RT_SECTION(6);
							if (ph == PH_NODE) { // the path goes on: its walk and the bounce arm of SHADE, in the same iteration
								walk_closest_pending();
								RT_SECTION(7);
								do_shade(std::integral_constant<int, 2>{});
							}
						}
						RT_SECTION(8);
					} else {
						load_pair_scene();
						if (ph == PH_LIGHT)
							do_light(L, sray);
						RT_SECTION(4);
						walk_shadow_pending(L, sray);
						RT_SECTION(5);
						if (ph == PH_SCATTER)
							do_scatter(L, sray);
						RT_SECTION(6);
	#if RT_PQ_SPLIT
						// the scattered ray's closest walk and the bounce arm of SHADE ride in the same iteration, so
						// the other super-phase only ever holds primary lanes and neither arm runs half empty
						walk_closest_pending();
						RT_SECTION(7);
						if (ph == PH_SHADE)
							do_shade(std::integral_constant<int, 2>{});
						RT_SECTION(8);
	#endif
					}
				}
	#if RT_PQ_SPLIT
				if (METHOD == 1) {
					// With the two super-phases above no walk is ever pending at the end of an iteration: the ray and
					// the walk state are dead here.  Saying so keeps them out of the loop-carried registers.
					ray.o = ray.d = ray.inv = ray.shear = v3s(0.0f);
					best_t = 0.0f;
					best_prim = kNoPrim;
					node = kRefDone;
					sp = 0;
					any_hit = false;
				}
	#endif
			} else {
				alive = __ballot(ph == PH_NEED_PIXEL) != 0ull;
				if (XCHG && METHOD == 1) // parked work left?  (a later push comes from a wave that is still alive and will drain it itself)
					alive = alive || __hip_atomic_load(&pool[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u ||
					        __hip_atomic_load(&pool[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
			}
		}
	}
#ifdef RT_STATS
	atomicAdd(&g_stats[20], st_lane_nodes); // lane-level node steps and primitive tests of the fine schedule
	atomicAdd(&g_stats[21], st_lane_prims);
	atomicMax(&g_stats[22], st_lane_maxsp);
	if (lane == 0u) {
		for (int k = 0; k < 9; ++k)
			atomicAdd(&g_stats[40 + k], st_sect[k]);
		if (!FINE) // (the fine schedule's clocks live in [50..57])
			for (int k = 9; k < 16; ++k)
				atomicAdd(&g_stats[40 + k], st_sect[k]);
		for (int k = 0; k < 8; ++k)
			atomicAdd(&g_stats[24 + k], st_hist[k]);
		for (int k = 0; k <= PH_COUNT; ++k)
			atomicAdd(&g_stats[50 + k], st_fine_clock[k]);
		atomicAdd(&g_stats[0], st_iters[0]); atomicAdd(&g_stats[1], st_active[0]);
		atomicAdd(&g_stats[2], st_iters[1]); atomicAdd(&g_stats[3], st_active[1]);
		atomicAdd(&g_stats[4], st_gen);
		for (int k = 0; k < PH_COUNT; ++k) {
			atomicAdd(&g_stats[5 + k], st_fine_iters[k]);
			atomicAdd(&g_stats[5 + PH_COUNT + k], st_fine_active[k]);
		}
	}
#endif
	// ---- SamplerProgress.rays_shot: wave reduction, one atomic per wave ----
	unsigned long long *const rays_shot = K->rays_shot;
	if (rays_shot != nullptr) {
		for (int off = 32; off > 0; off >>= 1)
			rays_total += __shfl_down(rays_total, off);
		if (lane == 0u)
			atomicAdd(rays_shot, rays_total);
	}
}

// ---- sample_split > 1: add the per-chunk sums of every pixel, in chunk order, and divide by spp:
// (sum_c sum_c) / spp in f32 (the definition in rt_hip.h) ----
__global__ __launch_bounds__(256) void combine_chunks_kernel(const DevRenderParams P, const float *__restrict__ partial, float *__restrict__ out)
{
	const uint32_t wp = blockIdx.x * blockDim.x + threadIdx.x;
	if (wp >= P.n_work)
		return;
	uint32_t x, y;
	if (!work_to_pixel(P, wp, x, y))
		return; // edge-tile padding
	V3 acc = v3s(0.0f);
	for (uint32_t c = 0; c < P.sample_split; ++c) {
		const float *m = partial + 3u * ((size_t)c * P.n_work + wp);
		acc.x = acc.x + m[0];
		acc.y = acc.y + m[1];
		acc.z = acc.z + m[2];
	}
	const size_t o = P.shard_layout ? (size_t)wp : (size_t)y * P.width + x;
	out[3u * o + 0u] = acc.x / (float)P.spp;
	out[3u * o + 1u] = acc.y / (float)P.spp;
	out[3u * o + 2u] = acc.z / (float)P.spp;
}

// Zeroes the per-launch counters (and, for packed / multi-shard outputs, the output buffer) with a kernel
// instead of hipMemsetAsync: memset nodes of a captured HIP graph were observed to replay an 8-byte
// memset with a garbage fill value on ROCm 7.2 (tests/test_gpu_parity.py::test_render_device_is_graph_capturable),
// and one launch replaces up to three.
__global__ __launch_bounds__(256) void reset_kernel(uint32_t *__restrict__ work_counter, unsigned long long *__restrict__ rays_shot,
                                                    float *__restrict__ out, size_t n_out_floats)
{
	const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) {
		*work_counter = 0u;
		if (rays_shot)
			*rays_shot = 0ull;
	}
	for (size_t k = i; k < n_out_floats; k += (size_t)gridDim.x * blockDim.x)
		out[k] = 0.0f;
}
hipError_t launch_reset(hipStream_t stream, uint32_t *work_counter, unsigned long long *rays_shot, float *out, size_t n_out_floats)
{
	size_t blocks = (n_out_floats + 255) / 256;
	if (blocks < 1)
		blocks = 1;
	if (blocks > 4096)
		blocks = 4096;
	hipLaunchKernelGGL(reset_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, work_counter, rays_shot, out, n_out_floats);
	return hipGetLastError();
}

// ---- output stage on the device: save_data_to_image's `(val.powf(1.0 / gamma) * 255.999) as u8`
// (crates/output/src/lib.rs:89-97) applied where the frame lives, so a caller that wants an 8-bit image copies
// W*H*3 bytes over PCIe instead of W*H*12.  Streaming: 4 values per lane, one dwordx4 load and one dword store. ----
__global__ __launch_bounds__(256) void quantise_kernel(const float *__restrict__ rgb, size_t n_values, float inv_gamma, uint8_t *__restrict__ out)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	const size_t n4 = n_values / 4;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
		const float4 v = reinterpret_cast<const float4 *>(rgb)[i];
		const uint32_t q = (uint32_t)rt_quantise_u8(v.x, inv_gamma) | ((uint32_t)rt_quantise_u8(v.y, inv_gamma) << 8) |
		                   ((uint32_t)rt_quantise_u8(v.z, inv_gamma) << 16) | ((uint32_t)rt_quantise_u8(v.w, inv_gamma) << 24);
		reinterpret_cast<uint32_t *>(out)[i] = q;
	}
	if (blockIdx.x == 0 && threadIdx.x < (n_values & 3))
		out[n4 * 4 + threadIdx.x] = rt_quantise_u8(rgb[n4 * 4 + threadIdx.x], inv_gamma);
}
// ... and for caller-supplied pointers that are not 16-byte (input) / 4-byte (output) aligned, e.g. a view into a larger
// buffer at an odd offset: one value per lane
__global__ __launch_bounds__(256) void quantise_unaligned_kernel(const float *__restrict__ rgb, size_t n_values, float inv_gamma, uint8_t *__restrict__ out)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride)
		out[i] = rt_quantise_u8(rgb[i], inv_gamma);
}
hipError_t launch_quantise(hipStream_t stream, const float *rgb, size_t n_values, float inv_gamma, uint8_t *out)
{
	if ((reinterpret_cast<uintptr_t>(rgb) & 15u) != 0u || (reinterpret_cast<uintptr_t>(out) & 3u) != 0u) {
		size_t blocks = (n_values + 255) / 256;
		blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
		hipLaunchKernelGGL(quantise_unaligned_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, rgb, n_values, inv_gamma, out);
		return hipGetLastError();
	}
	size_t blocks = (n_values / 4 + 255) / 256;
	if (blocks < 1)
		blocks = 1;
	if (blocks > 8192)
		blocks = 8192;
	hipLaunchKernelGGL(quantise_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, rgb, n_values, inv_gamma, out);
	return hipGetLastError();
}

// ---- multi-device scenes (rt_scene_create_multi): a member's packed shard (RT_LAYOUT_SHARD: its tiles in work order) is
// written into the frame on the gathering device.  The index map is the render kernel's own work_to_pixel. ----
__global__ __launch_bounds__(256) void scatter_shard_kernel(const DevRenderParams P, const float *__restrict__ shard, float *__restrict__ frame)
{
	const uint32_t wp = blockIdx.x * blockDim.x + threadIdx.x;
	if (wp >= P.n_work)
		return;
	uint32_t x, y;
	if (!work_to_pixel(P, wp, x, y))
		return; // edge-tile padding
	const size_t o = (size_t)y * P.width + x;
	frame[3u * o + 0u] = shard[3u * (size_t)wp + 0u];
	frame[3u * o + 1u] = shard[3u * (size_t)wp + 1u];
	frame[3u * o + 2u] = shard[3u * (size_t)wp + 2u];
}
hipError_t launch_scatter_shard(hipStream_t stream, const DevRenderParams &P, const float *shard, float *frame)
{
	hipLaunchKernelGGL(scatter_shard_kernel, dim3((P.n_work + 255u) / 256u), dim3(256), 0, stream, P, shard, frame);
	return hipGetLastError();
}
// SamplerProgress.rays_shot of a multi-device render: the members' counters, summed on the gathering device
__global__ void sum_u64_kernel(const unsigned long long *__restrict__ parts, uint32_t n, unsigned long long *__restrict__ out)
{
	unsigned long long t = 0;
	for (uint32_t i = 0; i < n; ++i)
		t += parts[i];
	*out = t;
}
hipError_t launch_sum_u64(hipStream_t stream, const unsigned long long *parts, uint32_t n, unsigned long long *out)
{
	hipLaunchKernelGGL(sum_u64_kernel, dim3(1), dim3(1), 0, stream, parts, n, out);
	return hipGetLastError();
}

hipError_t launch_combine(hipStream_t stream, const DevRenderParams &P, const float *partial, float *out)
{
	hipLaunchKernelGGL(combine_chunks_kernel, dim3((P.n_work + 255u) / 256u), dim3(256), 0, stream, P, partial, out);
	return hipGetLastError();
}

// ---- batch hit queries (AccelerationStructure::check_hit / check_hit_index) ----
struct DevRayDesc {
	float origin[3], direction[3];
};
struct DevHitRecord {
	float t, point[3], error[3], normal[3], uv[2];
	int32_t has_uv, out;
	uint32_t material, found;
	unsigned long long index;
};
static_assert(sizeof(DevHitRecord) == 72, "must match rt_hit_record");

__device__ __forceinline__ void store_record(DevHitRecord &o, const Hit &h, uint32_t material, unsigned long long index, bool found)
{
	o.t = h.t;
	o.point[0] = h.point.x; o.point[1] = h.point.y; o.point[2] = h.point.z;
	o.error[0] = h.error.x; o.error[1] = h.error.y; o.error[2] = h.error.z;
	o.normal[0] = h.normal.x; o.normal[1] = h.normal.y; o.normal[2] = h.normal.z;
	o.uv[0] = h.uvx; o.uv[1] = h.uvy;
	o.has_uv = h.has_uv ? 1 : 0;
	o.out = h.out ? 1 : 0;
	o.material = mat_handle_index(material); // (the caller's index, not the handle the kernels carry)
	o.found = found ? 1u : 0u;
	o.index = index;
}

template <bool PRUNE>
__global__ __launch_bounds__(256) void check_hit_kernel(const DevScene S, const DevRayDesc *__restrict__ rays, uint64_t n,
                                                        DevHitRecord *__restrict__ outr)
{
	using F = FeatFull; // batch queries serve every scene: all primitive types compiled in
	extern __shared__ __align__(16) uint32_t lds[];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t *stk = lds + wave * (S.stack_depth * kStackStride) + lane;
	// the whole worst case in LDS: the overflow branch is never taken (its base only has to be some global pointer;
	// a literal null there sends this compiler's SimplifyCFG into a crash)
	const StackMem SM = {S.stack_depth, 0u, reinterpret_cast<uint32_t *>(outr), lds};
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const Ray r = ray_new<F>(v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]),
	                      v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]));
	float t;
	uint32_t prim;
	trace_closest<F, PRUNE>(S, S, SM, r, stk, t, prim);
	Hit h;
	uint32_t m;
	if (prim != kNoPrim) {
		make_hit<F>(S, prim, r, t, h, m);
		store_record(outr[i], h, m, prim, true);
	} else {
		make_sky_hit(S, h, m);
		store_record(outr[i], h, m, 0xFFFFFFFFFFFFFFFFull, true);
	}
}

template <bool PRUNE>
__global__ __launch_bounds__(256) void check_hit_index_kernel(const DevScene S, const DevRayDesc *__restrict__ rays,
                                                              const unsigned long long *__restrict__ object_index, uint64_t n,
                                                              DevHitRecord *__restrict__ outr)
{
	using F = FeatFull; // batch queries serve every scene: all primitive types compiled in
	extern __shared__ __align__(16) uint32_t lds[];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t *stk = lds + wave * (S.stack_depth * kStackStride) + lane;
	const StackMem SM = {S.stack_depth, 0u, reinterpret_cast<uint32_t *>(outr), lds}; // see check_hit_kernel
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const Ray r = ray_new<F>(v3(rays[i].origin[0], rays[i].origin[1], rays[i].origin[2]),
	                      v3(rays[i].direction[0], rays[i].direction[1], rays[i].direction[2]));
	const uint32_t index = (uint32_t)object_index[i];
	const PrimGeom g = load_prim<F>(S, index);
	Hit h;
	h.t = 0.0f;
	h.point = h.error = h.normal = v3s(0.0f);
	h.uvx = h.uvy = 0.0f;
	h.has_uv = h.out = false;
	uint32_t m = 0;
	bool found = false;
	float lt;
	if (prim_t<F>(g, r, lt) && lt > 0.0f) {
		if (!trace_any<F, PRUNE>(S, S, SM, r, stk, lt, index)) {
			make_hit<F>(S, index, r, lt, h, m);
			found = true;
		}
	}
	store_record(outr[i], h, m, object_index[i], found);
}

#ifdef RT_STATS
// ---- diagnostic build only: a register-lean TRAVERSAL-ONLY persistent kernel (tests/probes/gpu_trace_queue.py).
// Lanes pull rays from a queue, walk the wide tree (NODE / LEAF voted as in the fine schedule), write (t, primitive) and
// refill in place.  It holds nothing but the ray, the best hit and the stack, so it can run at up to 8 waves/SIMD: the
// experiment behind DESIGN.md section 8.1 (what a wavefront split could give the big-tree configurations). ----
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void trace_queue_kernel(const DevScene S, const DevRayDesc *__restrict__ rays, uint32_t n, float2 *__restrict__ out,
                                                                 uint32_t *__restrict__ counter, unsigned long long *__restrict__ steps_out,
                                                                 uint32_t stack_cap, uint32_t ovf_depth, uint32_t *__restrict__ ovf)
{
	using F = Feat<true, true, false, false>;
	extern __shared__ __align__(16) uint32_t lds[];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	StackMem SM;
	SM.cap = stack_cap;
	SM.ovf_depth = ovf_depth;
	SM.ovf = ovf;
	SM.region = lds;
	uint32_t *stk = lds + wave * (stack_cap * kStackStride) + lane;
	enum { EMPTY = 0, NODE = 1, LEAF = 2, DONE = 3 };
	int ph = EMPTY;
	Ray ray;
	ray.o = ray.d = ray.inv = ray.shear = v3s(0.0f);
	uint32_t node = kRefDone, best_prim = kNoPrim, id = 0;
	int sp = 0;
	float best_t = 0.0f;
	unsigned long long n_steps = 0;
	uint32_t wq_next = 0, wq_end = 0;
	auto finish = [&]() {
		out[id] = make_float2(best_t, __uint_as_float(best_prim));
		ph = EMPTY;
	};
	for (;;) {
		const unsigned long long need = __ballot(ph == EMPTY);
		if (need != 0ull) {
			const uint32_t cnt = (uint32_t)__popcll(need), avail = wq_end - wq_next;
			uint32_t base = wq_end;
			if (avail < cnt) {
				const int leader = __ffsll((long long)need) - 1;
				uint32_t claimed = 0;
				if ((int)lane == leader)
					claimed = atomicAdd(counter, 64u);
				base = __shfl(claimed, leader);
			}
			if (ph == EMPTY) {
				const uint32_t r = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
				id = r < avail ? wq_next + r : base + (r - avail);
				if (id >= n) {
					ph = DONE;
				} else {
					ray = ray_new<F>(v3(rays[id].origin[0], rays[id].origin[1], rays[id].origin[2]),
					                 v3(rays[id].direction[0], rays[id].direction[1], rays[id].direction[2]));
					best_t = 0.0f;
					best_prim = kNoPrim;
					sp = 0;
					node = S.root4_ref;
					ph = ref_is_leaf(node) ? LEAF : NODE;
				}
			}
			if (avail < cnt) {
				wq_next = base + (cnt - avail);
				wq_end = base + 64u;
			} else {
				wq_next += cnt;
			}
		}
		const uint32_t c_node = (uint32_t)__popcll(__ballot(ph == NODE)), c_leaf = (uint32_t)__popcll(__ballot(ph == LEAF));
		if (c_node + c_leaf == 0u) {
			if (__ballot(ph == EMPTY) == 0ull)
				break;
			continue;
		}
		if (c_leaf >= kDrainLanes || c_node == 0u) {
			if (ph == LEAF) {
				uint32_t leaf_ref;
				if (wide_leaf_hit(S, node, ray, leaf_ref))
					closest_in_leaf<F>(S, ray, leaf_ref, best_t, best_prim);
				if (sp == 0) {
					finish();
				} else {
					--sp;
					node = ovf_depth == 0u ? stack_load<false>(SM, stk, sp) : stack_load<true>(SM, stk, sp);
					ph = ref_is_leaf(node) ? LEAF : NODE;
				}
			}
		} else {
#pragma unroll 1
			for (int step = 0; step < kNodeStepsPerVote; ++step) {
				if (ph == NODE) {
					n_steps += 1;
					if (ovf_depth == 0u)
						node = descend4<true, false>(S, SM, ray, node, stk, sp, best_prim != kNoPrim, best_t);
					else
						node = descend4<true, true>(S, SM, ray, node, stk, sp, best_prim != kNoPrim, best_t);
					if (node == kRefDone)
						finish();
					else if (ref_is_leaf(node))
						ph = LEAF;
				}
			}
		}
	}
	for (int off = 32; off > 0; off >>= 1)
		n_steps += __shfl_down(n_steps, off);
	if (lane == 0u)
		atomicAdd(steps_out, n_steps);
}

hipError_t launch_trace_queue(int waves, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream, const DevScene &S, const void *rays, uint32_t n, void *out,
                              uint32_t *counter, unsigned long long *steps, uint32_t cap, uint32_t ovf_depth, uint32_t *ovf)
{
#define RT_TQ(W) \
	if (waves == W) { \
		hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void *>(trace_queue_kernel<W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
		if (e_ != hipSuccess) \
			return e_; \
		hipLaunchKernelGGL(trace_queue_kernel<W>, dim3(n_blocks), dim3(256), lds_bytes, stream, S, static_cast<const DevRayDesc *>(rays), n, \
		                   static_cast<float2 *>(out), counter, steps, cap, ovf_depth, ovf); \
		return hipGetLastError(); \
	}
	RT_TQ(3) RT_TQ(4) RT_TQ(5) RT_TQ(6) RT_TQ(8)
#undef RT_TQ
	return hipErrorInvalidValue;
}
#endif

// ---- rt_selftest_lean: the short arithmetic forms of rt_lean.h against the plain operators / rt_detmath.h, on the device.
// Operand classes: 0 lean_div, 1 lean_div_fix (numerator may be zero / inf / NaN), 2 lean_inv, 3 lean_div3 (shared reciprocal),
// 4 lean_sqrt, 5 lean_sincos, 6 lean_acos_dev, 7 lean_atan2, 8 ray_new (fast path and fallback against the plain operators).
// mismatches[k] counts results whose BITS differ (two NaNs count as equal). ----
__device__ __forceinline__ bool same_f32(float a, float b) { return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b); }
__device__ __forceinline__ float tame_from_bits(uint32_t u, int lo_exp, int hi_exp) // random sign and mantissa, exponent in [lo_exp, hi_exp]
{
	const uint32_t span = (uint32_t)(hi_exp - lo_exp + 1);
	const uint32_t e = (uint32_t)(127 + lo_exp) + ((u >> 23) & 0xFFu) % span;
	return __uint_as_float((u & 0x807FFFFFu) | (e << 23));
}
__device__ __noinline__ Ray ray_new_plain(V3 origin, V3 direction) // the plain operators, kept out of line so nothing is shared with the short form
{
	Ray r;
	direction = direction / mag(direction);
	r.o = origin;
	r.d = direction;
	r.inv = v3(1.0f / direction.x, 1.0f / direction.y, 1.0f / direction.z);
	const float ax = fabsf(direction.x), ay = fabsf(direction.y), az = fabsf(direction.z);
	const bool swap = (ax > ay && ax > az) || (ay > az);
	const float sx = swap ? direction.z : direction.x;
	const float sz = swap ? direction.x : direction.z;
	r.shear = v3(-sx / sz, -direction.y / sz, 1.0f / sz);
	return r;
}
__global__ __launch_bounds__(256) void selftest_lean_kernel(uint64_t n_per_thread, uint64_t seed, unsigned long long *__restrict__ mismatches)
{
	const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	rt_rng rng;
	rt_rng_seed(&rng, seed, tid, 0x5E1F7E57ull);
	unsigned long long bad[9] = {};
	for (uint64_t it = 0; it < n_per_thread; ++it) {
		const uint32_t u0 = rt_rng_u32(&rng), u1 = rt_rng_u32(&rng), u2 = rt_rng_u32(&rng), u3 = rt_rng_u32(&rng);
		// tame operands at the EDGES of what the call sites guarantee as well as in the middle
		const float d = tame_from_bits(u0, -81, 41), n = tame_from_bits(u1, -81, 41);
		if (fabsf(n) <= fabsf(d) * 0x1p95f && fabsf(d) <= fabsf(n) * 0x1p120f) {
			bad[0] += !same_f32(lean_div(n, d), n / d);
			bad[1] += !same_f32(lean_div_fix(n, d), n / d);
		}
		const uint32_t pick = u2 & 7u; // special numerators for the fix-up form
		const float ns = pick == 0u ? 0.0f : (pick == 1u ? -0.0f : (pick == 2u ? INFINITY : (pick == 3u ? -INFINITY : (pick == 4u ? __uint_as_float(0x7FC00000u) : n))));
		if (pick < 5u)
			bad[1] += !same_f32(lean_div_fix(ns, d), ns / d);
		bad[2] += !same_f32(lean_inv(d), 1.0f / d);
		{
			const V3 v = v3(tame_from_bits(u1, -60, 20), tame_from_bits(u2, -60, 20), tame_from_bits(u3, -60, 20));
			const float dd = tame_from_bits(u0, -20, 20);
			const V3 a = lean_div3(v, dd), b = v / dd;
			bad[3] += !(same_f32(a.x, b.x) && same_f32(a.y, b.y) && same_f32(a.z, b.z));
			const V3 vz = v3((u3 & 1u) ? 0.0f : v.x, (u3 & 2u) ? -0.0f : v.y, v.z);
			const V3 af = lean_div3_fix(vz, dd), bf = vz / dd;
			bad[3] += !(same_f32(af.x, bf.x) && same_f32(af.y, bf.y) && same_f32(af.z, bf.z));
		}
		{
			// square roots: the whole stated domain -- zero, [2^-96, inf], NaN, negatives -- and squares near exact roots
			const uint32_t k = u3 & 15u;
			float x = fabsf(tame_from_bits(u0, -96, 127));
			if (k == 0u) x = 0.0f; else if (k == 1u) x = -0.0f; else if (k == 2u) x = INFINITY; else if (k == 3u) x = -fabsf(n);
			else if (k == 4u) { const float r = fabsf(tame_from_bits(u1, -40, 40)); x = r * r; }
			else if (k == 5u) x = 1.0f - (float)(u1 >> 8) * 5.9604644775390625e-08f; // 1 - r, the Lambert sampler's argument
			else if (k == 6u) x = __uint_as_float(0x0F800000u + (u1 & 0xFFu));          // just above 2^-96
			bad[4] += !same_f32(lean_sqrt(x), sqrtf(x));
		}
		{
			const float r = (float)(u0 >> 8) * 5.9604644775390625e-08f;
			const float ang = (u1 & 1u) ? 2.0f * kPi * r : ((u1 & 2u) ? kPi * r * (1.0f + 0x1p-20f) : (float)(int32_t)(u2 >> 9) * r - 4194304.0f * r);
			float s_, c_;
			lean_sincos(ang, s_, c_);
			bad[5] += !(same_f32(s_, rt_sinf(ang)) && same_f32(c_, rt_cosf(ang)));
		}
		{
			const uint32_t k = u3 & 7u;
			float x = 2.0f * ((float)(u0 >> 8) * 5.9604644775390625e-08f) - 1.0f;
			if (k == 0u) x = tame_from_bits(u0, -30, 1); else if (k == 1u) x = (u1 & 1u) ? 1.0f : -1.0f; else if (k == 2u) x = __uint_as_float(0x3F000000u + (u1 & 3u) - 1u);
			else if (k == 3u) x = __uint_as_float(0x7FC00000u);
			bad[6] += !same_f32(lean_acos_dev(x), rt_acosf(x));
		}
		{
			const uint32_t k = u3 >> 28;
			float y = tame_from_bits(u0, -30, 30), x = tame_from_bits(u1, -30, 30);
			if (k == 0u) y = 0.0f; else if (k == 1u) x = -0.0f; else if (k == 2u) { x = 0.0f; y = -0.0f; } else if (k == 3u) y = (u2 & 1u) ? x : -x;
			else if (k == 4u) x = INFINITY; else if (k == 5u) { x = -INFINITY; y = INFINITY; } else if (k == 6u) y = __uint_as_float(0x7FC00000u);
			else if (k == 7u) { x = __uint_as_float(u0); y = __uint_as_float(u1); }
			bad[7] += !same_f32(lean_atan2(y, x), rt_atan2f(y, x));
		}
		{
			const uint32_t k = u3 & 15u;
			V3 dir = v3(tame_from_bits(u0, -8, 8), tame_from_bits(u1, -8, 8), tame_from_bits(u2, -8, 8));
			if (k == 0u) dir.x = 0.0f; else if (k == 1u) dir = v3(tame_from_bits(u0, -62, -58), tame_from_bits(u1, -8, 8), tame_from_bits(u2, -22, 21));
			else if (k == 2u) dir = v3(tame_from_bits(u0, -70, 70), tame_from_bits(u1, -70, 70), tame_from_bits(u2, -70, 70));
			else if (k == 3u) dir.y = -0.0f;
			const Ray a = ray_new<FeatFull>(v3s(0.0f), dir), b = ray_new_plain(v3s(0.0f), dir);
			bad[8] += !(same_f32(a.d.x, b.d.x) && same_f32(a.d.y, b.d.y) && same_f32(a.d.z, b.d.z) && same_f32(a.inv.x, b.inv.x) && same_f32(a.inv.y, b.inv.y) &&
			            same_f32(a.inv.z, b.inv.z) && same_f32(a.shear.x, b.shear.x) && same_f32(a.shear.y, b.shear.y) && same_f32(a.shear.z, b.shear.z));
		}
	}
	for (int k = 0; k < 9; ++k)
		if (bad[k])
			atomicAdd(&mismatches[k], bad[k]);
}
hipError_t launch_selftest_lean(hipStream_t stream, uint32_t blocks, uint64_t n_per_thread, uint64_t seed, unsigned long long *mismatches)
{
	hipLaunchKernelGGL(selftest_lean_kernel, dim3(blocks), dim3(256), 0, stream, n_per_thread, seed, mismatches);
	return hipGetLastError();
}

// ---- launchers (called from rt_api.cpp) ----
#ifdef RT_STATS
extern "C" int rt_debug_hist(unsigned long long *out260, int reset)
{
	if (hipMemcpyFromSymbol(out260, HIP_SYMBOL(g_hist), sizeof(g_hist)) != hipSuccess)
		return -1;
	if (reset) {
		unsigned long long z[4 * 65] = {};
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_hist), z, sizeof z);
	}
	return 0;
}
extern "C" int rt_debug_stats(unsigned long long *out16, int reset)
{
	if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stats), sizeof(g_stats)) != hipSuccess)
		return -1;
	if (reset) {
		unsigned long long z[64] = {};
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof z);
	}
	return 0;
}
#endif

uint32_t render_block_threads(int feature_set, bool fine, bool xchg)
{
	if (fine && xchg)
		return 512u; // KernelShape<F, true, true>::block
	if (feature_set == 3)
		return (uint32_t)KernelShape<FeatPair>::block;
	if (feature_set == 0)
		return (uint32_t)KernelShape<Feat<false, false, false, false>>::block;
	if (feature_set == 1)
		return fine ? (uint32_t)KernelShape<Feat<true, true, false, false>, true>::block : (uint32_t)KernelShape<Feat<true, true, false, false>, false>::block;
	return 256u;
}

uint32_t render_max_block_threads(int feature_set, bool fine, bool xchg)
{
	if (fine || xchg)
		return render_block_threads(feature_set, fine, xchg);
	if (feature_set == 3)
		return (uint32_t)KernelShape<FeatPair>::max_block;
	if (feature_set == 0)
		return (uint32_t)KernelShape<Feat<false, false, false, false>>::max_block;
	if (feature_set == 1)
		return (uint32_t)KernelShape<Feat<true, true, false, false>>::max_block;
	return render_block_threads(feature_set, fine, xchg);
}

// waves per SIMD the kernel's register budget is declared for (its __launch_bounds__)
uint32_t render_waves_per_simd(int feature_set, bool fine)
{
	if (feature_set == 3)
		return (uint32_t)KernelShape<FeatPair>::waves_per_simd;
	if (feature_set == 0)
		return fine ? (uint32_t)KernelShape<Feat<false, false, false, false>, true>::waves_per_simd : (uint32_t)KernelShape<Feat<false, false, false, false>, false>::waves_per_simd;
	if (feature_set == 1)
		return fine ? (uint32_t)KernelShape<Feat<true, true, false, false>, true>::waves_per_simd : (uint32_t)KernelShape<Feat<true, true, false, false>, false>::waves_per_simd;
	return fine ? (uint32_t)KernelShape<FeatFull, true>::waves_per_simd : (uint32_t)KernelShape<FeatFull, false>::waves_per_simd;
}

size_t render_lds_bytes(const DevScene &S, bool sky_lds, bool scene_lds, uint32_t waves_per_block, uint32_t stack_cap)
{
	size_t words = scene_lds ? S.blob_bytes / 4u : 0u;
	if (sky_lds) {
		const uint32_t n_all = S.sky.res_y * (S.sky.res_x + 1u) + S.sky.res_y + 1u;
		words += (n_all + 3u) & ~3u;
		words += (S.sky.res_y + 1u) * S.sky.guide_k / 4u;
	}
	words += (size_t)waves_per_block * stack_cap * kStackStride;
	return words * sizeof(uint32_t);
}

typedef void (*render_fn)(const RenderArgs);

// feature sets the render kernel is instantiated for (rt_api.cpp picks the smallest that covers the scene)
using FeatSpheres = Feat<false, false, false, false>; // spheres, Lambertian/Emit, Solid/Lerp, sky is the only light (rtweekend1)
using FeatSimple = Feat<true, true, false, false>;    // + triangles and emissive primitives (overshadowed, the synthetic meshes)

template <class F> static render_fn pick_render_f(int method, bool prune, bool fine, bool sky_lds, bool xchg)
{
#ifdef RT_ONLY_HEADLINE // ISA studies (tests/probes/isa_headline.sh): instantiate the kernels of BASELINE configs 2 and 3 only
	if constexpr (!(F::cmat || F::ctex))
		if (method == 1 && !prune && !fine && sky_lds && !xchg)
			return render_kernel<1, false, false, true, F>;
#ifdef RT_HEADLINE_RUNNABLE // ... plus their twins without the sky tables in LDS, which the host asks about when it sizes a launch: a
	                        // library that can RUN configs 2 and 3 (tests/probes/gpu_r03_one_ab.sh), built in 25 s instead of 95
	if constexpr (!(F::cmat || F::ctex))
		if (method == 1 && !prune && !fine && !sky_lds && !xchg)
			return render_kernel<1, false, false, false, F>;
#endif
	return nullptr;
#else
	if constexpr (F::pair) { // the exhaustive coarse kernels only: that is where the host uses this set (rt_api.cpp)
		if (xchg || prune || fine)
			return nullptr;
		if (method == 0)
			return render_kernel<0, false, false, false, F>;
		return sky_lds ? render_kernel<1, false, false, true, F> : render_kernel<1, false, false, false, F>;
	} else {
	if (xchg) { // built for the coarse exhaustive MIS kernels (configs 2 and 3) and for the fine schedule
		if (method == 1 && !prune && !fine)
			return sky_lds ? render_kernel<1, false, false, true, F, true> : render_kernel<1, false, false, false, F, true>;
		if (fine && method == 0)
			return render_kernel<0, true, true, false, F, true>;
		if (fine && method == 1)
			return sky_lds ? render_kernel<1, true, true, true, F, true> : render_kernel<1, true, true, false, F, true>;
		return nullptr;
	}
#define RT_PICK(M, P, G, L) \
	if (method == M && prune == P && fine == G && sky_lds == L) \
		return render_kernel<M, P, G, L, F>;
	// (prune, fine): exhaustive+coarse for tiny trees, pruned+coarse for small ones, pruned+fine for big ones
	RT_PICK(0, false, false, false)
	RT_PICK(0, true, false, false)
	RT_PICK(0, true, true, false)
	RT_PICK(1, false, false, false)
	RT_PICK(1, false, false, true)
	RT_PICK(1, true, false, false)
	RT_PICK(1, true, false, true)
	RT_PICK(1, true, true, false)
	RT_PICK(1, true, true, true)
#undef RT_PICK
	return nullptr;
	}
#endif
}

// feature_set: 0 spheres-only, 1 simple, 2 full, 3 spheres-only over a two-leaf tree of single primitives (FeatPair)
static render_fn pick_render(int method, bool prune, bool fine, bool sky_lds, int feature_set, bool xchg = false)
{
	if (method == 0)
		sky_lds = false; // the naive integrator never touches the sky tables
	if (fine)
		prune = true; // the fine schedule is only built with the pruned walk
	if (feature_set == 0)
		return pick_render_f<FeatSpheres>(method, prune, fine, sky_lds, xchg);
	if (feature_set == 1)
		return pick_render_f<FeatSimple>(method, prune, fine, sky_lds, xchg);
	if (feature_set == 3)
		return pick_render_f<FeatPair>(method, prune, fine, sky_lds, xchg);
	return pick_render_f<FeatFull>(method, prune, fine, sky_lds, xchg);
}
bool render_exchange_available(int method, bool prune, bool fine, int feature_set) { (void)feature_set; return fine || (method == 1 && !prune); }
// LDS of a pool of `slots` parked paths + pixels (incl. the header and the slack of aligning it to 16 bytes)
size_t render_exchange_lds_bytes(uint32_t slots) { return (size_t)(4u + 4u + (kXchgBStride + kXchgPStride) * slots) * sizeof(uint32_t); }
uint32_t render_exchange_max_slots() { return kXchgMaxSlots; }
// ... and of the two record pools of the fine schedule (+ state words, + one scratch row per wave)
size_t render_exchange_fine_lds_bytes(uint32_t waves_per_block) { return (size_t)(4u + kXchgFineHdr + waves_per_block * 64u + 2u * kXchgFineSlots * kXchgRecWords) * sizeof(uint32_t); }

hipError_t render_occupancy(int method, bool prune, bool fine, bool sky_lds, int feature_set, size_t lds_bytes, int *blocks_per_cu, bool xchg,
                            uint32_t block_threads)
{
	render_fn fn = pick_render(method, prune, fine, sky_lds, feature_set, xchg);
	if (!fn)
		return hipErrorInvalidValue;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess)
		return e;
	return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, fn, block_threads ? block_threads : render_block_threads(feature_set, fine, xchg), lds_bytes);
}

hipError_t launch_render(int method, bool prune, bool fine, bool sky_lds, int feature_set, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream,
                         const DevScene &S, const DevCamera &cam, const DevRenderParams &P, float *out,
                         unsigned long long *rays_shot, uint32_t *work_counter, uint32_t *stack_ovf, bool xchg, const DevPairScene *pair,
                         uint32_t block_threads)
{
	render_fn fn = pick_render(method, prune, fine, sky_lds, feature_set, xchg);
	if (!fn)
		return hipErrorInvalidValue;
	RenderArgs A;
	A.S = S;
	A.cam = cam;
	A.P = P;
	A.out = out;
	A.rays_shot = rays_shot;
	A.work_counter = work_counter;
	A.stack_ovf = stack_ovf;
	A.pair = pair ? *pair : DevPairScene{};
	hipLaunchKernelGGL(fn, dim3(n_blocks), dim3(block_threads ? block_threads : render_block_threads(feature_set, fine, xchg)), lds_bytes, stream, A);
	return hipGetLastError();
}

hipError_t launch_check_hit(bool prune, hipStream_t stream, const DevScene &S, const void *rays, uint64_t n, void *out)
{
	const size_t lds_bytes = (size_t)4 * S.stack_depth * kStackStride * sizeof(uint32_t);
	const uint32_t blocks = (uint32_t)((n + 255) / 256);
	// deep trees: more than the default 64 KB of dynamic LDS per workgroup (the whole worst-case stack lives in LDS here)
	hipError_t e = hipFuncSetAttribute(prune ? reinterpret_cast<const void *>(check_hit_kernel<true>) : reinterpret_cast<const void *>(check_hit_kernel<false>),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess)
		return e;
	if (prune)
		hipLaunchKernelGGL(check_hit_kernel<true>, dim3(blocks), dim3(256), lds_bytes, stream, S,
		                   static_cast<const DevRayDesc *>(rays), n, static_cast<DevHitRecord *>(out));
	else
		hipLaunchKernelGGL(check_hit_kernel<false>, dim3(blocks), dim3(256), lds_bytes, stream, S,
		                   static_cast<const DevRayDesc *>(rays), n, static_cast<DevHitRecord *>(out));
	return hipGetLastError();
}

hipError_t launch_check_hit_index(bool prune, hipStream_t stream, const DevScene &S, const void *rays, const void *object_index,
                                  uint64_t n, void *out)
{
	const size_t lds_bytes = (size_t)4 * S.stack_depth * kStackStride * sizeof(uint32_t);
	const uint32_t blocks = (uint32_t)((n + 255) / 256);
	hipError_t e = hipFuncSetAttribute(prune ? reinterpret_cast<const void *>(check_hit_index_kernel<true>)
	                                         : reinterpret_cast<const void *>(check_hit_index_kernel<false>),
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
	if (e != hipSuccess)
		return e;
	if (prune)
		hipLaunchKernelGGL(check_hit_index_kernel<true>, dim3(blocks), dim3(256), lds_bytes, stream, S,
		                   static_cast<const DevRayDesc *>(rays), static_cast<const unsigned long long *>(object_index), n,
		                   static_cast<DevHitRecord *>(out));
	else
		hipLaunchKernelGGL(check_hit_index_kernel<false>, dim3(blocks), dim3(256), lds_bytes, stream, S,
		                   static_cast<const DevRayDesc *>(rays), static_cast<const unsigned long long *>(object_index), n,
		                   static_cast<DevHitRecord *>(out));
	return hipGetLastError();
}

} // namespace rt
