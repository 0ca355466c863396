// rt_api.cpp -- the extern "C" boundary of librt_hip.so (include/rt_hip.h).
//
// Owns device memory for one scene (rt_scene), turns RenderOptions-style arguments into one
// persistent kernel launch (rt_render.hip) and reports errors as codes + rt_last_error().
// There is NO CPU fallback: without a HIP device every compute entry point fails with
// RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rt_hip.h"
#include "../../include/rt_detmath.h"
#include <dlfcn.h>
#include "rt_build.h"
#include "rt_types.h"

namespace rt {
size_t render_lds_bytes(const DevScene &S, bool sky_lds, bool scene_lds, uint32_t waves_per_block, uint32_t stack_cap);
uint32_t render_waves_per_simd(int feature_set, bool fine);
uint32_t render_block_threads(int feature_set, bool fine, bool xchg = false);
size_t render_exchange_fine_lds_bytes(uint32_t waves_per_block);
uint32_t render_max_block_threads(int feature_set, bool fine, bool xchg);
hipError_t render_occupancy(int method, bool prune, bool fine, bool sky_lds, int feature_set, size_t lds_bytes, int *blocks_per_cu, bool xchg,
                            uint32_t block_threads = 0);
bool render_exchange_available(int method, bool prune, bool fine, int feature_set);
size_t render_exchange_lds_bytes(uint32_t slots);
uint32_t render_exchange_max_slots();
hipError_t launch_render(int method, bool prune, bool fine, bool sky_lds, int feature_set, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream,
                         const DevScene &S, const DevCamera &cam, const DevRenderParams &P, float *out,
                         unsigned long long *rays_shot, uint32_t *work_counter, uint32_t *stack_ovf, bool xchg, const DevPairScene *pair,
                         uint32_t block_threads = 0);
hipError_t launch_combine(hipStream_t stream, const DevRenderParams &P, const float *partial, float *out);
hipError_t launch_reset(hipStream_t stream, uint32_t *work_counter, unsigned long long *rays_shot, float *out, size_t n_out_floats);
hipError_t launch_quantise(hipStream_t stream, const float *rgb, size_t n_values, float inv_gamma, uint8_t *out);
hipError_t launch_check_hit(bool prune, hipStream_t stream, const DevScene &S, const void *rays, uint64_t n, void *out);
#ifdef RT_STATS
hipError_t launch_trace_queue(int waves, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream, const DevScene &S, const void *rays, uint32_t n, void *out,
                              uint32_t *counter, unsigned long long *steps, uint32_t cap, uint32_t ovf_depth, uint32_t *ovf);
#endif
hipError_t launch_scatter_shard(hipStream_t stream, const DevRenderParams &P, const float *shard, float *frame);
hipError_t launch_sum_u64(hipStream_t stream, const unsigned long long *parts, uint32_t n, unsigned long long *out);
hipError_t launch_selftest_lean(hipStream_t stream, uint32_t blocks, uint64_t n_per_thread, uint64_t seed, unsigned long long *mismatches);
hipError_t launch_check_hit_index(bool prune, hipStream_t stream, const DevScene &S, const void *rays, const void *object_index,
                                  uint64_t n, void *out);
} // namespace rt

using namespace rt;

static thread_local std::string g_error;

static int fail(int code, const std::string &msg)
{
	g_error = msg;
	return code;
}
static int hip_fail(hipError_t e, const char *what)
{
	g_error = std::string(what) + ": " + hipGetErrorString(e);
	return e == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP;
}
#define HIP_TRY(expr)                       \
	do {                                    \
		hipError_t e_ = (expr);             \
		if (e_ != hipSuccess)               \
			return hip_fail(e_, #expr);     \
	} while (0)

// Measured crossovers (tests/probes/gpu_crossover_probe.py, tests/probes/gpu_crossover_mesh.py, 1080p):
//   exhaustive walk + coarse schedule wins up to ~100 primitives, the pruned walk beyond;
//   the fine schedule (every step voted) wins from a few thousand triangles (10 k: 36 -> 26 ms, 1 M: 189 -> 77 ms)
//   but only ties on sphere-only scenes even at 16 k (96 vs 106 ms): sphere leaves are cheap, triangle leaves
//   are long and divergent.
constexpr uint32_t kPruneAbove = 100;
constexpr uint32_t kFineAboveTriangles = 4096; // round 2, with the wide walk under both schedules (8 spp 1080p, MIS coarse / fine):
                                               // 2 000 triangles 11.9 / 13.4 ms, 4 000: 14.9 / 16.2, 10 000: 30.2 / 26.2 (naive crosses near 3 000)
constexpr uint32_t kFineAboveSpheres = 32768;

struct rt_scene {
	int device = 0;
	HostScene host;
	DevScene dev{};
	std::vector<void *> allocations;
	hipStream_t stream = nullptr; // used by the blocking entry points
	uint32_t *d_work_counter = nullptr;
	unsigned long long *d_rays = nullptr;
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	bool timed = false;
	uint32_t n_launches = 0;
	int n_cus = 0;
	int traversal_mode = -1; // -1 auto, 0 exhaustive (reference order of work), 1 pruned
	int schedule_mode = -1;  // -1 auto, 0 coarse (two super-phases), 1 fine (every step voted)
	int feature_set = 2;     // smallest kernel variant covering the scene: 0 spheres-only, 1 simple, 2 full
	int min_feature_set = 0; // what the scene needs (feature_set may be forced larger for tests)
	// the tree is one inner node over two leaves of one primitive each: launches that would run the spheres-only exhaustive
	// coarse kernels run their FeatPair twins (rt_types.h) -- unless a feature set was asked for by name (RT_TUNE_FEATURE_SET)
	bool pair_tree = false, feature_set_forced = false;
	DevPairScene pair{}; // pair_tree: the scene as the FeatPair kernels take it, in their kernel arguments (rt_types.h)
	bool scene_lds_allowed = true;
	float *d_partial = nullptr; // sample_split > 1: per-chunk means, grown on demand
	size_t partial_floats = 0;
	// rt_sample_image: two batches in flight (device + pinned host buffers, copy stream, events)
	float *d_prog[2] = {nullptr, nullptr};
	float *h_prog[2] = {nullptr, nullptr};
	unsigned long long *d_prog_rays = nullptr; // [2]
	unsigned long long *h_prog_rays = nullptr; // [2], pinned
	size_t d_prog_floats[2] = {0, 0}, h_prog_floats[2] = {0, 0};
	hipStream_t copy_stream = nullptr;
	hipEvent_t ev_batch[2] = {nullptr, nullptr}, ev_copy[2] = {nullptr, nullptr};
	size_t max_lds = 65536;
	rt_launch_info last_launch{};
	uint32_t stack_cap_override = 0; // RT_TUNE_STACK_CAP
	int exchange_mode = 0;           // RT_TUNE_EXCHANGE
	uint32_t stack_depth_narrow = 2; // HostScene::stack_depth_narrow (members of a multi-device scene have no host scene of their own)
	uint32_t *d_stack_ovf = nullptr; // traversal-stack overflow area (deep trees under the fine schedule), grown on demand
	size_t stack_ovf_words = 0;
	uint8_t *d_rgb8 = nullptr; // rt_render_rgb8: the quantised frame
	size_t d_rgb8_bytes = 0;
	// ---- multi-device scenes (rt_scene_create_multi).  The handle a caller holds is the HEAD: an ordinary scene on
	// devices[0] that additionally owns one member scene per further device (uploaded from the head's host build) and
	// gathers their tile shards into its own frames.  Members render like any single-device scene. ----
	std::vector<rt_scene *> peers;         // head only: the members on devices[1..n-1]
	bool member_call = false;              // set while the head renders its own shard through the single-device path
	float *d_shard = nullptr;              // every member incl. the head: its packed shard (RT_LAYOUT_SHARD)
	size_t shard_floats = 0;
	unsigned long long *d_member_rays = nullptr; // the member's own ray counter (the head's d_rays holds the job's total)
	hipEvent_t ev_shard = nullptr;         // member: its shard is rendered
	hipEvent_t ev_begin = nullptr;         // head: the caller's stream has reached this render
	float *d_gather = nullptr;             // head: the peers' shards, once gathered
	size_t gather_floats = 0;
	unsigned long long *d_gather_rays = nullptr; // head: [n] the members' ray counters
	void *nccl_comms = nullptr;            // head: ncclComm_t[n] when the devices are distinct and RCCL is usable
	int gather_mode = 0;                   // rt_gather_mode, decided when the scene is created (rt_scene_create_multi)
	std::string gather_note;               // why (rt_scene_gather_info)
	hipEvent_t ev_gathered = nullptr;      // head: the last render's gather + scatter have read every member's shard
	bool gathered_once = false;
};

template <class T> static int upload(rt_scene *s, const T *src, size_t count, const T **dst)
{
	void *p = nullptr;
	const size_t bytes = (count ? count : 1) * sizeof(T);
	HIP_TRY(hipMalloc(&p, bytes));
	s->allocations.push_back(p);
	if (count)
		HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
	*dst = static_cast<const T *>(p);
	return RT_OK;
}

// ---- RCCL, loaded when a multi-device scene over DISTINCT devices is created.  The gather of a multi-device render is the one
// collective of the path: grouped ncclSend / ncclRecv of the members' shards into the head's device over
// xGMI.  The library is not a link-time dependency.  Look-up order: the file RT_HIP_RCCL_LIB names (tests: a stand-in that checks
// the call pattern, tests/cpp/fake_rccl.cpp); a copy ALREADY LOADED into the process (RTLD_NOLOAD: a process that carries
// PyTorch's librccl must not get a second one, with its own bootstrap threads and its own view of the devices); then the
// system's.  If none is usable, or ncclCommInitAll refuses the device list, the same bytes move with hipMemcpyPeerAsync. ----
struct RcclApi {
	void *lib = nullptr;
	int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
	int (*CommDestroy)(void *comm) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*Send)(const void *buf, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
	int (*Recv)(void *buf, size_t count, int datatype, int peer, void *comm, hipStream_t stream) = nullptr;
	bool ok = false;
	std::string origin; // which file, and how it was found (rt_scene_gather_info)
};
static RcclApi load_rccl()
{
	RcclApi a;
	const char *forced = std::getenv("RT_HIP_RCCL_LIB");
	if (forced && *forced) {
		a.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
		a.origin = std::string(forced) + (a.lib ? " (RT_HIP_RCCL_LIB)" : " (RT_HIP_RCCL_LIB: cannot be loaded)");
	} else {
		static const char *const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
		for (const char *name : names) { // a copy the process already carries first
			a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
			if (a.lib) {
				a.origin = std::string(name) + " (already loaded in this process)";
				break;
			}
		}
		for (const char *name : names) {
			if (a.lib)
				break;
			a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (a.lib)
				a.origin = std::string(name) + " (loaded on demand)";
		}
		if (!a.lib)
			a.origin = "librccl not found";
	}
	if (!a.lib)
		return a;
	a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(a.lib, "ncclCommInitAll"));
	a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
	a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(a.lib, "ncclGroupStart"));
	a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(a.lib, "ncclGroupEnd"));
	a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(a.lib, "ncclSend"));
	a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(a.lib, "ncclRecv"));
	a.ok = a.CommInitAll && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Send && a.Recv;
	if (!a.ok)
		a.origin += ": lacks one of ncclCommInitAll / ncclCommDestroy / ncclGroupStart / ncclGroupEnd / ncclSend / ncclRecv";
	return a;
}
static RcclApi &rccl()
{
	// (looked up once per process -- except under RT_HIP_RCCL_LIB, which tests change between scenes)
	static RcclApi api;
	static std::string key = "\x01";
	const char *forced = std::getenv("RT_HIP_RCCL_LIB");
	const std::string now = forced ? forced : "";
	if (key != now) {
		api = load_rccl();
		key = now;
	}
	return api;
}
constexpr int kNcclFloat32 = 7; // ncclFloat32 of rccl.h

extern "C" void rt_scene_destroy(rt_scene *s);
// what a multi-device head owns beyond an ordinary scene
static void multi_release(rt_scene *s)
{
	if (s->nccl_comms) {
		void **comms = static_cast<void **>(s->nccl_comms);
		for (size_t i = 0; i < s->peers.size() + 1; ++i)
			if (comms[i])
				(void)rccl().CommDestroy(comms[i]);
		delete[] comms;
		s->nccl_comms = nullptr;
	}
	for (rt_scene *m : s->peers)
		rt_scene_destroy(m);
	s->peers.clear();
	if (s->device != RT_DEVICE_NONE) {
		(void)hipSetDevice(s->device);
		if (s->d_shard) (void)hipFree(s->d_shard);
		if (s->d_gather) (void)hipFree(s->d_gather);
		if (s->d_gather_rays) (void)hipFree(s->d_gather_rays);
		if (s->d_member_rays) (void)hipFree(s->d_member_rays);
		if (s->ev_shard) (void)hipEventDestroy(s->ev_shard);
		if (s->ev_begin) (void)hipEventDestroy(s->ev_begin);
		if (s->ev_gathered) (void)hipEventDestroy(s->ev_gathered);
	}
	s->d_shard = s->d_gather = nullptr;
	s->d_gather_rays = s->d_member_rays = nullptr;
	s->ev_shard = s->ev_begin = s->ev_gathered = nullptr;
}

// The transport of a multi-device scene's gather (rt_gather_mode in rt_hip.h), decided when the scene is created:
//   members that share the head's device         a plain device-to-device copy
//   distinct devices, RCCL usable                 ncclCommInitAll over the device list -> grouped ncclSend / ncclRecv
//   otherwise (RT_HIP_NO_RCCL, no library, a library that lacks a symbol, ncclCommInitAll refusing the list)
//                                                 hipMemcpyPeerAsync, with peer access enabled head <- member where the devices allow
//                                                 it (RT_GATHER_PEER) and staged through the host by the runtime where they do not
//                                                 (RT_GATHER_PEER_STAGED: correct, slower; the note says which pair)
// A failure to ENABLE peer access that the device pair reports as possible is an error (the caller would silently get the slow path).
static int multi_decide_gather(rt_scene *head, const std::vector<rt_scene *> &members)
{
	const uint32_t n = (uint32_t)members.size();
	bool distinct = true, all_same = true;
	for (uint32_t a = 0; a < n; ++a) {
		all_same = all_same && members[a]->device == head->device;
		for (uint32_t b = a + 1; b < n; ++b)
			distinct = distinct && members[a]->device != members[b]->device;
	}
	head->gather_mode = RT_GATHER_NONE;
	head->gather_note.clear();
	if (n <= 1)
		return RT_OK;
	// RT_HIP_TEST_RCCL_ANY_LIST (tests only): offer RCCL a list with repeated devices too -- a real RCCL refuses it, which is the
	// refusal path; the stand-in of tests/cpp/fake_rccl.cpp accepts it, which runs the grouped send / recv call pattern on one GPU
	const bool test_any = std::getenv("RT_HIP_TEST_RCCL_ANY_LIST") != nullptr;
	if (all_same && !test_any) {
		head->gather_mode = RT_GATHER_SAME_DEVICE;
		head->gather_note = "every member shares the head's device: device-to-device copies";
		return RT_OK;
	}
	const bool offer = distinct || test_any;
	std::string why;
	if (std::getenv("RT_HIP_NO_RCCL") != nullptr)
		why = "RT_HIP_NO_RCCL is set";
	else if (!offer)
		why = "the device list repeats a device (RCCL wants distinct devices)";
	else if (!rccl().ok)
		why = "RCCL unusable: " + rccl().origin;
	else {
		void **comms = new void *[n]();
		std::vector<int> devs(n);
		for (uint32_t m = 0; m < n; ++m)
			devs[m] = members[m]->device;
		const int rc = rccl().CommInitAll(comms, (int)n, devs.data());
		(void)hipSetDevice(head->device); // (ncclCommInitAll visits every device)
		if (rc == 0) {
			head->nccl_comms = comms;
			head->gather_mode = RT_GATHER_RCCL;
			head->gather_note = "grouped ncclSend / ncclRecv, " + rccl().origin;
			return RT_OK;
		}
		delete[] comms;
		why = "ncclCommInitAll refused the device list (ncclResult " + std::to_string(rc) + "), " + rccl().origin;
	}
	// peer copies into the head's HBM: enable head <- member access where the pair supports it
	head->gather_mode = RT_GATHER_PEER;
	std::string staged;
	for (uint32_t m = 1; m < n; ++m) {
		if (members[m]->device == head->device)
			continue;
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, head->device, members[m]->device) != hipSuccess)
			can = 0;
		if (can) {
			if (hipSetDevice(head->device) != hipSuccess)
				return fail(RT_ERR_HIP, "hipSetDevice failed");
			const hipError_t e = hipDeviceEnablePeerAccess(members[m]->device, 0);
			if (e == hipErrorPeerAccessAlreadyEnabled)
				(void)hipGetLastError(); // (another scene, or the caller, enabled it: fine)
			else if (e != hipSuccess)
				return fail(RT_ERR_HIP, "multi-device scene: device " + std::to_string(head->device) + " can access device " + std::to_string(members[m]->device) +
				                            " as a peer but hipDeviceEnablePeerAccess failed: " + hipGetErrorString(e));
		} else {
			head->gather_mode = RT_GATHER_PEER_STAGED;
			staged += (staged.empty() ? "" : ", ") + std::to_string(members[m]->device);
		}
	}
	head->gather_note = "hipMemcpyPeerAsync (" + why + ")";
	if (all_same) {
		head->gather_mode = RT_GATHER_SAME_DEVICE;
		head->gather_note = "every member shares the head's device: device-to-device copies (" + why + ")";
	}
	if (!staged.empty())
		head->gather_note += "; no peer access from device " + std::to_string(head->device) + " to device(s) " + staged + ": those copies are staged through the host";
	(void)hipSetDevice(head->device);
	return RT_OK;
}

// Lays a built scene out in the HBM of s->device (every array of rt_types.h) and creates the scene's stream, events and
// counters.  `h` is only read: the members of a multi-device scene (rt_scene_create_multi) are uploaded from one host build.
static int upload_scene(rt_scene *s, const HostScene &h)
{
	int rc = RT_OK;
	auto bail = [&](int code) { return code; }; // (the caller destroys the scene)
	const int device = s->device;
	if (hipSetDevice(device) != hipSuccess)
		return bail(fail(RT_ERR_HIP, "hipSetDevice failed"));
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) != hipSuccess)
		return bail(fail(RT_ERR_HIP, "hipGetDeviceProperties failed"));
	// the texture records and the tiny-scene blob carry device pointers: per-device copies
	std::vector<DevTexture> textures = h.textures;
	std::vector<uint32_t> blob;
	uint32_t blob_off[8] = {};
	s->n_cus = prop.multiProcessorCount;
	s->max_lds = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : 65536;

	DevScene &D = s->dev;
	std::memset(&D, 0, sizeof D);
	// per-texture payloads first, so the texture records can point at them
	for (size_t i = 0; i < h.textures.size(); ++i) {
		if (!h.tex_images[i].empty()) {
			if ((rc = upload(s, h.tex_images[i].data(), h.tex_images[i].size(), &textures[i].image)) != RT_OK)
				return bail(rc);
		}
		if (!h.tex_perlin_vecs[i].empty()) {
			if ((rc = upload(s, h.tex_perlin_vecs[i].data(), h.tex_perlin_vecs[i].size(), &textures[i].perlin_vecs)) != RT_OK)
				return bail(rc);
			if ((rc = upload(s, h.tex_perlin_perm[i].data(), h.tex_perlin_perm[i].size(), &textures[i].perlin_perm)) != RT_OK)
				return bail(rc);
		}
	}
	if ((rc = upload(s, h.dev_nodes.data(), h.dev_nodes.size(), &D.nodes)) != RT_OK) return bail(rc);
	if (!h.dev_nodes4.empty()) { // hipMalloc aligns far beyond the 128 bytes a DevNode4 line needs
		// (the compact form: rt_types.h; rt_scene_get_wide_nodes / rt_scene_get_leaf_boxes show the explicit one)
		if ((rc = upload(s, h.dev_nodes4c.data(), h.dev_nodes4c.size(), &D.nodes4)) != RT_OK) return bail(rc);
		if ((rc = upload(s, h.leaf_box_c.data(), h.leaf_box_c.size(), &D.leaf_box)) != RT_OK) return bail(rc);
	}
	if ((rc = upload(s, h.dev_prims.data(), h.dev_prims.size(), &D.prims)) != RT_OK) return bail(rc);
	if ((rc = upload(s, h.dev_shade.data(), h.dev_shade.size(), &D.shade)) != RT_OK) return bail(rc);
	if ((rc = upload(s, h.prim_rank.data(), h.prim_rank.size(), &D.prim_rank)) != RT_OK) return bail(rc);
	{
		const uint32_t *d_big = nullptr;
		if ((rc = upload(s, h.big_leaves.data(), h.big_leaves.size(), &d_big)) != RT_OK) return bail(rc);
		D.big_leaves = reinterpret_cast<const uint2 *>(d_big);
	}
	if ((rc = upload(s, h.materials.data(), h.materials.size(), &D.materials)) != RT_OK) return bail(rc);
	if ((rc = upload(s, textures.data(), textures.size(), &D.textures)) != RT_OK) return bail(rc);
	if ((rc = upload(s, h.dev_lights.data(), h.dev_lights.size(), &D.lights)) != RT_OK) return bail(rc);
	const float *d_sky = nullptr;
	if ((rc = upload(s, h.sky_cdf.data(), h.sky_cdf.size(), &d_sky)) != RT_OK) return bail(rc);
	const uint8_t *d_guide = nullptr;
	if ((rc = upload(s, h.sky_guide.data(), h.sky_guide.size(), &d_guide)) != RT_OK) return bail(rc);
	{
		// tiny scenes: one packed copy of every array, staged into LDS by the render kernel.  Built here
		// (not in rt_build.cpp) because the texture records must already hold their device pointers.
		auto pad16 = [](size_t n) { return (n + 15) & ~(size_t)15; };
		const size_t sizes[8] = {h.dev_nodes.size() * sizeof(DevNode),      h.dev_prims.size() * sizeof(DevPrim),
		                         h.dev_shade.size() * sizeof(DevShade),     h.prim_rank.size() * 4,
		                         h.materials.size() * sizeof(DevMaterial),  textures.size() * sizeof(DevTexture),
		                         h.dev_lights.size() * 4,                   h.big_leaves.size() * 4};
		const void *srcs[8] = {h.dev_nodes.data(), h.dev_prims.data(), h.dev_shade.data(), h.prim_rank.data(),
		                       h.materials.data(), textures.data(),  h.dev_lights.data(), h.big_leaves.data()};
		size_t total = 0;
		for (int i = 0; i < 8; ++i) {
			blob_off[i] = (uint32_t)total;
			total += pad16(sizes[i]);
		}
		if (total <= 12 * 1024) {
			blob.assign(total / 4, 0u);
			for (int i = 0; i < 8; ++i)
				if (sizes[i])
					std::memcpy(reinterpret_cast<char *>(blob.data()) + blob_off[i], srcs[i], sizes[i]);
			if ((rc = upload(s, blob.data(), blob.size(), &D.blob)) != RT_OK) return bail(rc);
			D.blob_bytes = (uint32_t)total;
			D.off_nodes = blob_off[0]; D.off_prims = blob_off[1]; D.off_shade = blob_off[2]; D.off_rank = blob_off[3];
			D.off_materials = blob_off[4]; D.off_textures = blob_off[5]; D.off_lights = blob_off[6]; D.off_big_leaves = blob_off[7];
		}
	}

	D.n_nodes = (uint32_t)h.dev_nodes.size();
	D.n_prims = (uint32_t)h.dev_prims.size();
	D.n_lights = (uint32_t)h.dev_lights.size();
	D.n_materials = (uint32_t)h.materials.size();
	D.n_textures = (uint32_t)h.textures.size();
	D.root_ref = h.root_ref;
	D.root4_ref = h.root4_ref;
	D.n_nodes4 = (uint32_t)h.dev_nodes4.size();
	std::memcpy(D.root_min, h.root_min, sizeof D.root_min);
	std::memcpy(D.root_max, h.root_max, sizeof D.root_max);
	D.stack_depth = h.stack_depth;
	s->stack_depth_narrow = h.stack_depth_narrow;
	D.has_triangles = h.has_triangles ? 1u : 0u;
	D.single_light = h.dev_lights.size() == 1 ? h.dev_lights[0] : kNoPrim;
	D.sky.texture = h.sky.texture;
	D.sky.material = mat_handle_make(h.sky.material, h.materials[h.sky.material].type, h.materials[h.sky.material].tex_type);
	D.sky.res_x = h.sky.sampler_res_x;
	D.sky.res_y = h.sky.sampler_res_y;
	D.sky.row_cdf = d_sky;
	D.sky.marginal_cdf = d_sky + (size_t)h.sky.sampler_res_y * (h.sky.sampler_res_x + 1u);
	D.sky.guide = h.sky_guide_k ? d_guide : nullptr;
	D.sky.guide_k = h.sky_guide_k;
	{ // divisions by launch constants (rt_build.h verified_reciprocal; rt_lean.h div_by_verified)
		float rc_pi = 0.0f, rc_tau = 0.0f;
		if (!verified_reciprocal(RT_PI, &rc_pi) || !verified_reciprocal(RT_TAU, &rc_tau) || rc_pi != 1.0f / RT_PI || rc_tau != 1.0f / RT_TAU)
			return bail(fail(RT_ERR_UNSUPPORTED, "self-check failed: division by pi / 2 pi through their reciprocals is not exact on this host (the kernels assume it)"));
		D.sky.inv_res_ok = 0u;
		D.sky.inv_res_x = D.sky.inv_res_y = 0.0f;
		if (h.sky.sampler_res_x != 0 && h.sky.sampler_res_y != 0 && h.sky.sampler_res_x < (1u << 24) && h.sky.sampler_res_y < (1u << 24)) {
			const bool ok_x = verified_reciprocal((float)h.sky.sampler_res_x, &D.sky.inv_res_x), ok_y = verified_reciprocal((float)h.sky.sampler_res_y, &D.sky.inv_res_y);
			const bool ok = ok_x && ok_y;
			D.sky.inv_res_ok = ok ? 1u : 0u;
		}
		// Lambertian numerators (rt_shade.h cosine_is_tame_): SolidColour textures, |colour x albedo| components zero or in
		// [2^-30, 2^30], vertex normals finite and below 2^20
		bool tame = true;
		for (const DevMaterial &m : h.materials) {
			if (m.type != RT_MAT_LAMBERTIAN)
				continue;
			tame = tame && m.tex_type == RT_TEX_SOLID;
			for (int k = 0; k < 3; ++k) {
				const float pr = std::fabs(m.tex_c1[k] * m.param);
				tame = tame && (pr == 0.0f || (pr >= 0x1p-30f && pr <= 0x1p30f)); // false for NaN
			}
		}
		for (const DevShade &sh : h.dev_shade)
			for (int k = 0; k < 3; ++k)
				tame = tame && std::fabs(sh.n0[k]) <= 0x1p20f && std::fabs(sh.n1[k]) <= 0x1p20f && std::fabs(sh.n2[k]) <= 0x1p20f;
		D.lambert_tame = tame ? 1u : 0u;
	}

	void *p = nullptr;
	if (hipMalloc(&p, sizeof(uint32_t)) != hipSuccess) return bail(fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc work counter"));
	s->allocations.push_back(p);
	s->d_work_counter = static_cast<uint32_t *>(p);
	if (hipMalloc(&p, sizeof(unsigned long long)) != hipSuccess) return bail(fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc rays"));
	s->allocations.push_back(p);
	s->d_rays = static_cast<unsigned long long *>(p);
	if (hipStreamCreate(&s->stream) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipStreamCreate failed"));
	if (hipEventCreate(&s->ev_start) != hipSuccess || hipEventCreate(&s->ev_stop) != hipSuccess)
		return bail(fail(RT_ERR_HIP, "hipEventCreate failed"));
	{ // which kernel variant covers this scene (rt_types.h Feat)
		bool cmat = false, ctex = false;
		for (const DevMaterial &m : h.materials)
			cmat = cmat || (m.type != RT_MAT_EMIT && m.type != RT_MAT_LAMBERTIAN);
		for (const DevTexture &t : textures)
			ctex = ctex || (t.type != RT_TEX_SOLID && t.type != RT_TEX_LERP);
		if (cmat || ctex)
			s->feature_set = 2;
		else if (h.has_triangles || !h.lights.empty())
			s->feature_set = 1;
		else
			s->feature_set = 0;
		s->min_feature_set = s->feature_set;
		if (h.dev_nodes.size() == 1 && (h.root_ref & kLeafFlag) == 0u) {
			// FeatPair (rt_types.h): one node over two single-primitive leaves, both primitives Lambertian over a SolidColour,
			// the sky's material an Emit (its texture is a SolidColour or a Lerp: no ctex in this feature set)
			const DevNode &n0 = h.dev_nodes[0];
			s->pair_tree = (n0.c0 & kLeafFlag) && (n0.c1 & kLeafFlag) && ((n0.c0 >> 26) & 31u) == 1u && ((n0.c1 >> 26) & 31u) == 1u;
			for (const DevPrim &q : h.dev_prims) {
				uint32_t meta;
				std::memcpy(&meta, &q.a[3], sizeof meta);
				const DevMaterial &m = h.materials[mat_handle_index(meta >> 2)];
				s->pair_tree = s->pair_tree && m.type == RT_MAT_LAMBERTIAN && m.tex_type == RT_TEX_SOLID;
			}
			s->pair_tree = s->pair_tree && h.materials[h.sky.material].type == RT_MAT_EMIT;
			s->pair_tree = s->pair_tree && D.lambert_tame != 0u; // (rt_shade.h cosine_is_tame_: these kernels do not read the flag)
			// ... and the hit record's (p - c) / r goes through the radius' reciprocal without asking (rt_intersect.h
			// make_sphere_hit_by_reciprocal): both must have passed the host's enumeration (rt_build.cpp, DevPrim::b[1])
			for (const DevPrim &q : h.dev_prims)
				s->pair_tree = s->pair_tree && q.b[1] != 0.0f;
			if (s->pair_tree) { // ... and that scene as kernel arguments, copied from the very records the other kernels read
				DevPairScene &q = s->pair;
				std::memcpy(q.c0min, n0.c0min, sizeof q.c0min); std::memcpy(q.c0max, n0.c0max, sizeof q.c0max);
				std::memcpy(q.c1min, n0.c1min, sizeof q.c1min); std::memcpy(q.c1max, n0.c1max, sizeof q.c1max);
				q.slot0 = n0.c0 & kLeafSlotMask;
				q.slot1 = n0.c1 & kLeafSlotMask;
				q.rank0 = h.prim_rank[q.slot0];
				q.rank1 = h.prim_rank[q.slot1];
				const uint32_t slots[2] = {q.slot0, q.slot1};
				for (int k = 0; k < 2; ++k) {
					const DevPrim &pr = h.dev_prims[slots[k]];
					uint32_t meta;
					std::memcpy(&meta, &pr.a[3], sizeof meta);
					const DevMaterial &m = h.materials[mat_handle_index(meta >> 2)];
					q.sphere[k][0] = pr.a[0]; q.sphere[k][1] = pr.a[1]; q.sphere[k][2] = pr.a[2]; q.sphere[k][3] = pr.b[0];
					q.lambert[k][0] = m.tex_c1[0]; q.lambert[k][1] = m.tex_c1[1]; q.lambert[k][2] = m.tex_c1[2]; q.lambert[k][3] = m.param;
				}
				q.inv_radius[0] = h.dev_prims[q.slot0].b[1];
				q.inv_radius[1] = h.dev_prims[q.slot1].b[1];
				const DevMaterial &sky = h.materials[h.sky.material];
				q.sky_param = sky.param;
				q.sky_tex_type = sky.tex_type;
				std::memcpy(q.sky_c1, sky.tex_c1, sizeof q.sky_c1);
				std::memcpy(q.sky_c2, sky.tex_c2, sizeof q.sky_c2);
			}
		}
	}
	if (const char *e = std::getenv("RT_HIP_TRAVERSAL")) { // "exhaustive" | "pruned": override the automatic choice
		if (std::strcmp(e, "exhaustive") == 0) s->traversal_mode = 0;
		if (std::strcmp(e, "pruned") == 0) s->traversal_mode = 1;
	}
	return RT_OK;
}

extern "C" {

const char *rt_last_error(void) { return g_error.c_str(); }
uint32_t rt_abi_version(void) { return RT_ABI_VERSION; }

int rt_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess)
		return 0;
	return n;
}

void rt_render_opts_default(rt_render_opts *o)
{
	if (!o)
		return;
	std::memset(o, 0, sizeof *o);
	o->width = 1920; // RenderOptions::default  samplers/mod.rs:31-41
	o->height = 1080;
	o->samples_per_pixel = 128;
	o->render_method = RT_METHOD_MIS;
	o->max_depth = 50;   // integrators/mod.rs:7
	o->rr_threshold = 3; // integrators/mod.rs:8
	o->shard_count = 1;
	o->output_layout = RT_LAYOUT_FRAME;
	o->sample_split = 1;
}

int rt_camera_new(rt_camera *out, const float origin[3], const float lookat[3], const float vup[3], float fov_degrees,
                  float aspect_ratio, float aperture, float focus_dist)
{
	if (!out || !origin || !lookat || !vup)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	camera_new(out, origin, lookat, vup, fov_degrees, aspect_ratio, aperture, focus_dist);
	return RT_OK;
}

void rt_scene_destroy(rt_scene *s)
{
	if (!s)
		return;
	if (s->device == RT_DEVICE_NONE) { // host-only scene: nothing lives on a GPU
		delete s;
		return;
	}
	multi_release(s);
	(void)hipSetDevice(s->device);
	for (void *p : s->allocations)
		(void)hipFree(p);
	if (s->d_partial)
		(void)hipFree(s->d_partial);
	if (s->d_rgb8)
		(void)hipFree(s->d_rgb8);
	if (s->d_stack_ovf)
		(void)hipFree(s->d_stack_ovf);
	for (int b = 0; b < 2; ++b) {
		if (s->d_prog[b]) (void)hipFree(s->d_prog[b]);
		if (s->h_prog[b]) (void)hipHostFree(s->h_prog[b]);
		if (s->ev_batch[b]) (void)hipEventDestroy(s->ev_batch[b]);
		if (s->ev_copy[b]) (void)hipEventDestroy(s->ev_copy[b]);
	}
	if (s->d_prog_rays) (void)hipFree(s->d_prog_rays);
	if (s->h_prog_rays) (void)hipHostFree(s->h_prog_rays);
	if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
	if (s->ev_start)
		(void)hipEventDestroy(s->ev_start);
	if (s->ev_stop)
		(void)hipEventDestroy(s->ev_stop);
	if (s->stream)
		(void)hipStreamDestroy(s->stream);
	delete s;
}

int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out)
{
	if (!desc || !out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	*out = nullptr;
	if (device == RT_DEVICE_NONE) { // Bvh::new only: the tree, the primitive order and the lights, no GPU touched
		rt_scene *hs = new rt_scene();
		hs->device = RT_DEVICE_NONE;
		std::string herr;
		const int hrc = build_host_scene(desc, hs->host, herr);
		if (hrc != RT_OK) {
			delete hs;
			return fail(hrc, herr);
		}
		*out = hs;
		return RT_OK;
	}
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
		return fail(RT_ERR_NO_DEVICE, "no HIP device: the rt_hip back end has no CPU fallback");
	if (device < 0 || device >= n_dev)
		return fail(RT_ERR_INVALID_ARGUMENT, "device index out of range");

	rt_scene *s = new rt_scene();
	s->device = device;
	std::string err;
	int rc = build_host_scene(desc, s->host, err);
	if (rc != RT_OK) {
		delete s;
		return fail(rc, err);
	}
	rc = upload_scene(s, s->host);
	if (rc != RT_OK) {
		rt_scene_destroy(s);
		return rc;
	}
	*out = s;
	return RT_OK;
}

int rt_scene_create_multi(const rt_scene_desc *desc, const int *devices, uint32_t n_devices, rt_scene **out)
{
	if (!desc || !out || !devices || n_devices == 0)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument or empty device list");
	*out = nullptr;
	if (n_devices > 64)
		return fail(RT_ERR_INVALID_ARGUMENT, "more than 64 devices");
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0)
		return fail(RT_ERR_NO_DEVICE, "no HIP device: the rt_hip back end has no CPU fallback");
	for (uint32_t i = 0; i < n_devices; ++i)
		if (devices[i] < 0 || devices[i] >= n_dev)
			return fail(RT_ERR_INVALID_ARGUMENT, "device index out of range");
	rt_scene *head = nullptr;
	int rc = rt_scene_create(desc, devices[0], &head); // Bvh::new once, on the host
	if (rc != RT_OK)
		return rc;
	for (uint32_t i = 1; i < n_devices; ++i) { // the same arrays in the HBM of every further device
		rt_scene *m = new rt_scene();
		m->device = devices[i];
		head->peers.push_back(m);
		rc = upload_scene(m, head->host);
		if (rc != RT_OK) {
			rt_scene_destroy(head);
			return rc;
		}
	}
	std::vector<rt_scene *> members{head};
	members.insert(members.end(), head->peers.begin(), head->peers.end());
	for (rt_scene *m : members) {
		if (hipSetDevice(m->device) != hipSuccess || hipEventCreateWithFlags(&m->ev_shard, hipEventDisableTiming) != hipSuccess ||
		    hipMalloc(reinterpret_cast<void **>(&m->d_member_rays), sizeof(unsigned long long)) != hipSuccess) {
			rt_scene_destroy(head);
			return fail(RT_ERR_HIP, "multi-device scene: event / counter allocation failed");
		}
	}
	if (hipSetDevice(head->device) != hipSuccess || hipEventCreateWithFlags(&head->ev_begin, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&head->ev_gathered, hipEventDisableTiming) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&head->d_gather_rays), n_devices * sizeof(unsigned long long)) != hipSuccess) {
		rt_scene_destroy(head);
		return fail(RT_ERR_HIP, "multi-device scene: event / counter allocation failed");
	}
	// ---- how the members' shards will reach the head: decided HERE, once, so that no render ever initialises a communicator,
	// loads a library or changes a device's peer mappings (rt_render_device stays free of host synchronisation from its first
	// call on, and graph-capturable) ----
	rc = multi_decide_gather(head, members);
	if (rc != RT_OK) {
		rt_scene_destroy(head);
		return rc;
	}
	*out = head;
	return RT_OK;
}

int rt_rccl_probe(int *usable, char *note, uint64_t note_capacity)
{
	if (!usable)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	const RcclApi &a = rccl();
	*usable = (a.ok && std::getenv("RT_HIP_NO_RCCL") == nullptr) ? 1 : 0;
	if (note && note_capacity) {
		const std::string text = std::getenv("RT_HIP_NO_RCCL") != nullptr ? std::string("RT_HIP_NO_RCCL is set") : a.origin;
		std::strncpy(note, text.c_str(), (size_t)note_capacity - 1);
		note[note_capacity - 1] = 0;
	}
	return RT_OK;
}

int rt_scene_gather_info(const rt_scene *s, int *mode, char *note, uint64_t note_capacity)
{
	if (!s || !mode)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	*mode = s->peers.empty() ? RT_GATHER_NONE : s->gather_mode;
	if (note && note_capacity) {
		std::strncpy(note, s->gather_note.c_str(), (size_t)note_capacity - 1);
		note[note_capacity - 1] = 0;
	}
	return RT_OK;
}

int rt_scene_device_count(const rt_scene *s, uint32_t *n_devices)
{
	if (!s || !n_devices)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	*n_devices = s->device == RT_DEVICE_NONE ? 0u : (uint32_t)(1 + s->peers.size());
	return RT_OK;
}

int rt_scene_set_traversal(rt_scene *s, int mode)
{
	if (!s || mode < -1 || mode > 1)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	s->traversal_mode = mode;
	for (rt_scene *m : s->peers)
		m->traversal_mode = mode;
	return RT_OK;
}

static int set_tuning_one(rt_scene *s, int key, int value);
int rt_scene_set_tuning(rt_scene *s, int key, int value)
{
	if (!s)
		return fail(RT_ERR_INVALID_ARGUMENT, "null scene");
	const int rc = set_tuning_one(s, key, value);
	for (rt_scene *m : s->peers) // the members of a multi-device scene run the same kernels
		if (rc == RT_OK) {
			(void)set_tuning_one(m, key, value);
		}
	return rc;
}
static int set_tuning_one(rt_scene *s, int key, int value)
{
	switch (key) {
	case RT_TUNE_TRAVERSAL:
		return rt_scene_set_traversal(s, value);
	case RT_TUNE_FEATURE_SET: // may only grow: a smaller variant would lack code the scene needs
		if (value < s->min_feature_set || value > 2)
			return fail(RT_ERR_INVALID_ARGUMENT, "feature set must be between the scene's own and 2");
		s->feature_set = value;
		s->feature_set_forced = true;
		return RT_OK;
	case RT_TUNE_SCHEDULE:
		if (value < -1 || value > 1)
			return fail(RT_ERR_INVALID_ARGUMENT, "schedule must be -1, 0 or 1");
		s->schedule_mode = value;
		return RT_OK;
	case RT_TUNE_SCENE_IN_LDS:
		s->scene_lds_allowed = value != 0;
		return RT_OK;
	case RT_TUNE_STACK_CAP:
		if (value < 0 || value > 96)
			return fail(RT_ERR_INVALID_ARGUMENT, "stack cap must be 0 (automatic) or 1..96 entries");
		s->stack_cap_override = (uint32_t)value;
		return RT_OK;
	case RT_TUNE_EXCHANGE:
		if (value < 0 || value > 1)
			return fail(RT_ERR_INVALID_ARGUMENT, "exchange must be 0 (off) or 1 (on where the kernel has it)");
		s->exchange_mode = value;
		return RT_OK;
	case RT_TUNE_WALK:
		if (value < 0 || value > 1)
			return fail(RT_ERR_INVALID_ARGUMENT, "walk must be 0 (automatic) or 1 (two-child walk for every ray)");
		s->dev.narrow_only = (uint32_t)value;
		return RT_OK;
	default:
		return fail(RT_ERR_INVALID_ARGUMENT, "unknown tuning key");
	}
}

int rt_scene_counts(const rt_scene *s, uint64_t *n_nodes, uint64_t *n_primitives, uint64_t *n_lights)
{
	if (!s)
		return fail(RT_ERR_INVALID_ARGUMENT, "null scene");
	if (n_nodes) *n_nodes = s->host.nodes.size();
	if (n_primitives) *n_primitives = s->host.primitive_order.size();
	if (n_lights) *n_lights = s->host.lights.size();
	return RT_OK;
}
int rt_scene_get_nodes(const rt_scene *s, rt_bvh_node *out, uint64_t capacity)
{
	if (!s || !out || capacity < s->host.nodes.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	for (size_t i = 0; i < s->host.nodes.size(); ++i) {
		const HostNode &n = s->host.nodes[i];
		std::memcpy(out[i].min, n.min, 12);
		std::memcpy(out[i].max, n.max, 12);
		out[i].children[0] = n.child[0];
		out[i].children[1] = n.child[1];
		out[i].primitive_offset = n.primitive_offset;
		out[i].number_primitives = n.number_primitives;
	}
	return RT_OK;
}
int rt_scene_get_primitive_order(const rt_scene *s, uint64_t *out, uint64_t capacity)
{
	if (!s || !out || capacity < s->host.primitive_order.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	std::memcpy(out, s->host.primitive_order.data(), s->host.primitive_order.size() * sizeof(uint64_t));
	return RT_OK;
}
int rt_scene_wide_info(const rt_scene *s, uint64_t *n_wide_nodes, uint32_t *root_ref, uint32_t *stack_depth)
{
	if (!s)
		return fail(RT_ERR_INVALID_ARGUMENT, "null scene");
	if (n_wide_nodes) *n_wide_nodes = s->host.dev_nodes4.size();
	if (root_ref) *root_ref = s->host.root4_ref;
	if (stack_depth) *stack_depth = s->host.stack_depth;
	return RT_OK;
}
int rt_scene_get_wide_nodes(const rt_scene *s, void *out, uint64_t capacity_nodes)
{
	if (!s || !out || capacity_nodes < s->host.dev_nodes4.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	std::memcpy(out, s->host.dev_nodes4.data(), s->host.dev_nodes4.size() * sizeof(DevNodeQ4));
	return RT_OK;
}
int rt_scene_get_leaf_boxes(const rt_scene *s, float *out, uint64_t capacity_slots)
{
	if (!s || !out || capacity_slots < s->host.leaf_box.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	std::memcpy(out, s->host.leaf_box.data(), s->host.leaf_box.size() * sizeof(DevLeafBox));
	return RT_OK;
}
int rt_scene_get_wide_nodes_compact(const rt_scene *s, void *out, uint64_t capacity_nodes)
{
	if (!s || !out || capacity_nodes < s->host.dev_nodes4c.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	std::memcpy(out, s->host.dev_nodes4c.data(), s->host.dev_nodes4c.size() * sizeof(DevNodeQ4));
	return RT_OK;
}
int rt_scene_get_leaf_boxes_compact(const rt_scene *s, float *out, uint64_t capacity_leaves, uint64_t *n_leaves)
{
	if (!s)
		return fail(RT_ERR_INVALID_ARGUMENT, "null scene");
	if (n_leaves)
		*n_leaves = s->host.leaf_box_c.size();
	if (!out)
		return n_leaves ? RT_OK : fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	if (capacity_leaves < s->host.leaf_box_c.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	std::memcpy(out, s->host.leaf_box_c.data(), s->host.leaf_box_c.size() * sizeof(DevLeafBox));
	return RT_OK;
}
int rt_scene_get_lights(const rt_scene *s, uint64_t *out, uint64_t capacity)
{
	if (!s || !out || capacity < s->host.lights.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	std::memcpy(out, s->host.lights.data(), s->host.lights.size() * sizeof(uint64_t));
	return RT_OK;
}

} // extern "C"

// ---- render ----
namespace {

struct ShardGeometry {
	uint32_t tile_w, tile_h, tiles_x, tiles_y;
	uint64_t n_tiles_owned, n_work;
};

int shard_geometry(const rt_render_opts *o, ShardGeometry &g)
{
	if (!o)
		return fail(RT_ERR_INVALID_ARGUMENT, "null options");
	if (o->width < 2 || o->height < 2)
		return fail(RT_ERR_INVALID_ARGUMENT, "width and height must be >= 2 (u and v divide by W-1 and H-1)");
	if (o->width * o->height >= (1ull << 31))
		return fail(RT_ERR_UNSUPPORTED, "image larger than 2^31 pixels");
	if (o->shard_count == 0 || o->shard_index >= o->shard_count)
		return fail(RT_ERR_INVALID_ARGUMENT, "shard_index must be < shard_count and shard_count >= 1");
	g.tile_w = o->tile_width ? o->tile_width : 8;
	g.tile_h = o->tile_height ? o->tile_height : 8;
	g.tiles_x = (uint32_t)((o->width + g.tile_w - 1) / g.tile_w);
	g.tiles_y = (uint32_t)((o->height + g.tile_h - 1) / g.tile_h);
	const uint64_t n_tiles = (uint64_t)g.tiles_x * g.tiles_y;
	g.n_tiles_owned = n_tiles > o->shard_index ? (n_tiles - o->shard_index + o->shard_count - 1) / o->shard_count : 0;
	g.n_work = g.n_tiles_owned * g.tile_w * g.tile_h;
	return RT_OK;
}

} // namespace

// sample_split = 0 (automatic), resolved.  ONE rule: the library (single- and multi-device scenes), bench.py and the tests all get
// it from here (rt_scene_auto_sample_split).
// A lane folds one work item at a time and an item is 1 / S of a pixel's passes: with whole pixels (S = 1) a device cannot use
// more lanes than it owns pixels, and even one 1080p frame on one GPU is only eight items per resident lane, whose last ones run
// while most of the chip has nothing left (7 % of config 2, 22 % of config 4).  S = the power of two that gives the device
// >= 64 items per resident lane, at most 64 (a claim of 64 items then stays inside one tile: rt_render.hip acquire_tiles), with
// chunks of at least 16 passes when the device holds the whole frame and at least 4 when the frame is sharded over `n_sharers`
// devices (a device's 1/8 share of config 4 -- 256 passes of 259 k pixels -- needs chunks of 4 passes to keep its lanes supplied:
// 146.0 / 140 / 134.5 ms at S = 16 / 32 / 64 on one GPU's share, ideal 124: profiles/r03k_mesh1m_share_splits.txt).  Measured on
// one GPU, config 2: whole frame 93.4 ms at S = 1, 86.9 at S = 16; a 1/8 share 27.9 ms at S = 1, 11.1 at S = 32 / 64
// (profiles/r03k_split_sweep.txt).
static uint32_t auto_sample_split(int n_cus, uint64_t frame_pixels, uint64_t spp, uint32_t n_sharers)
{
	if (n_sharers == 0u)
		n_sharers = 1u;
	const uint64_t lanes = (uint64_t)(n_cus > 0 ? n_cus : 256) * 1024u;
	const uint64_t min_chunk = n_sharers == 1u ? 16u : 4u;
	uint32_t split = 1;
	while ((frame_pixels / n_sharers) * split < 64 * lanes && split < 64u && split < spp / min_chunk)
		split *= 2;
	if (split > spp)
		split = (uint32_t)(spp ? spp : 1u);
	return split;
}

// Scene-owned frame buffers, grown on first use / larger frames only: rt_render needs one device frame,
// rt_sample_image two device frames, two pinned host frames, a copy stream and its events.
static int ensure_frame_buffers(rt_scene *s, uint64_t n_floats, bool progressive)
{
	const int n_dev = progressive ? 2 : 1;
	for (int b = 0; b < n_dev; ++b) {
		if (n_floats <= s->d_prog_floats[b])
			continue;
		if (s->d_prog[b]) (void)hipFree(s->d_prog[b]);
		s->d_prog[b] = nullptr;
		s->d_prog_floats[b] = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_prog[b]), n_floats * sizeof(float)));
		s->d_prog_floats[b] = n_floats;
	}
	if (!progressive)
		return RT_OK;
	if (!s->copy_stream) {
		HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
		for (int b = 0; b < 2; ++b) {
			HIP_TRY(hipEventCreateWithFlags(&s->ev_batch[b], hipEventDisableTiming));
			HIP_TRY(hipEventCreateWithFlags(&s->ev_copy[b], hipEventDisableTiming));
		}
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_prog_rays), 2 * sizeof(unsigned long long)));
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s->h_prog_rays), 2 * sizeof(unsigned long long), hipHostMallocDefault));
	}
	for (int b = 0; b < 2; ++b) {
		if (n_floats <= s->h_prog_floats[b])
			continue;
		if (s->h_prog[b]) (void)hipHostFree(s->h_prog[b]);
		s->h_prog[b] = nullptr;
		s->h_prog_floats[b] = 0;
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&s->h_prog[b]), n_floats * sizeof(float), hipHostMallocDefault));
		s->h_prog_floats[b] = n_floats;
	}
	return RT_OK;
}

extern "C" {

int rt_scene_auto_sample_split(const rt_scene *s, const rt_render_opts *o, uint32_t *split)
{
	if (!s || !o || !split)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (s->device == RT_DEVICE_NONE)
		return fail(RT_ERR_NO_DEVICE, "host-only scene (RT_DEVICE_NONE): the split depends on the device's size");
	if (o->shard_count == 0 || o->samples_per_pixel == 0)
		return fail(RT_ERR_INVALID_ARGUMENT, "shard_count and samples_per_pixel must be >= 1");
	const uint32_t members = (uint32_t)(1 + s->peers.size());
	*split = auto_sample_split(s->n_cus, o->width * o->height, o->samples_per_pixel, members > 1u ? members : o->shard_count);
	return RT_OK;
}

int rt_render_output_floats(const rt_render_opts *o, uint64_t *n_floats)
{
	ShardGeometry g;
	int rc = shard_geometry(o, g);
	if (rc != RT_OK)
		return rc;
	if (!n_floats)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	*n_floats = o->output_layout == RT_LAYOUT_SHARD ? g.n_work * 3 : o->width * o->height * 3;
	return RT_OK;
}

int rt_shard_pixel_order(const rt_render_opts *o, uint64_t *out, uint64_t capacity)
{
	ShardGeometry g;
	int rc = shard_geometry(o, g);
	if (rc != RT_OK)
		return rc;
	if (!out || capacity < g.n_work)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	const uint32_t tile_pixels = g.tile_w * g.tile_h;
	for (uint64_t w = 0; w < g.n_work; ++w) { // same mapping as work_to_pixel() in rt_render.hip
		const uint64_t k = w / tile_pixels, in = w % tile_pixels;
		const uint64_t tile = o->shard_index + k * o->shard_count;
		const uint64_t ty = tile / g.tiles_x, tx = tile % g.tiles_x;
		const uint64_t x = tx * g.tile_w + in % g.tile_w, y = ty * g.tile_h + in / g.tile_w;
		out[w] = (x < o->width && y < o->height) ? y * o->width + x : UINT64_MAX; // UINT64_MAX: edge-tile padding
	}
	return RT_OK;
}

// ---- a render on a multi-device scene (rt_scene_create_multi): `d_out_rgb` / `d_rays_shot` live on the HEAD's device and
// `hip_stream` is a stream of that device, exactly as for a single-device scene.  Tile t of the frame belongs to member
// t % n (the partition of DESIGN.md section 7); every member renders its tiles packed (RT_LAYOUT_SHARD) on its own
// stream, the head on the caller's; the peers' shards are gathered into the head's HBM -- grouped ncclSend / ncclRecv
// when the devices are distinct and RCCL is usable, hipMemcpyPeerAsync otherwise (same-device members: a plain copy) --
// and one small kernel per shard writes them into the frame.  Nothing synchronises with the host. ----
static int render_device_multi_enqueue(rt_scene *head, const std::vector<rt_scene *> &members, const rt_camera *camera, const rt_render_opts *o,
                                       float *d_out_rgb, uint64_t *d_rays_shot, hipStream_t stream);
static int render_device_multi(rt_scene *head, const rt_camera *camera, const rt_render_opts *o, float *d_out_rgb, uint64_t *d_rays_shot,
                               hipStream_t stream)
{
	if (o->shard_count != 1 || o->shard_index != 0 || o->output_layout != RT_LAYOUT_FRAME)
		return fail(RT_ERR_UNSUPPORTED, "a multi-device scene shards the frame over its own devices: shard_count must be 1 and the layout RT_LAYOUT_FRAME");
	std::vector<rt_scene *> members{head};
	members.insert(members.end(), head->peers.begin(), head->peers.end());
	const int rc = render_device_multi_enqueue(head, members, camera, o, d_out_rgb, d_rays_shot, stream);
	if (rc != RT_OK) {
		// An error part-way leaves work in flight on the members' streams and the current device wherever the loop stood.  Drain
		// the members (their next render must not race this one's leftovers), go back to the head's device, keep the first error.
		const std::string first = g_error;
		for (rt_scene *m : members) {
			m->member_call = false;
			if (m != head && hipSetDevice(m->device) == hipSuccess)
				(void)hipStreamSynchronize(m->stream);
		}
		(void)hipSetDevice(head->device);
		(void)hipGetLastError();
		g_error = first;
	}
	return rc;
}

static int render_device_multi_enqueue(rt_scene *head, const std::vector<rt_scene *> &members, const rt_camera *camera, const rt_render_opts *o,
                                       float *d_out_rgb, uint64_t *d_rays_shot, hipStream_t stream)
{
	const uint32_t n = (uint32_t)members.size();
	rt_render_opts om = *o;
	om.shard_count = n;
	om.output_layout = RT_LAYOUT_SHARD;
	if (om.sample_split == 0) // 0 = automatic; 1 = the reference's strictly sequential fold
		om.sample_split = auto_sample_split(head->n_cus, o->width * o->height, o->samples_per_pixel, n);

	std::vector<uint64_t> n_floats(n, 0), offset(n, 0);
	uint64_t gather_total = 0;
	for (uint32_t m = 0; m < n; ++m) {
		om.shard_index = m;
		int rc = rt_render_output_floats(&om, &n_floats[m]);
		if (rc != RT_OK)
			return rc;
		if (m > 0) {
			offset[m] = gather_total;
			gather_total += n_floats[m];
		}
	}
	HIP_TRY(hipSetDevice(head->device));
	if (gather_total > head->gather_floats) { // grown on first use / larger frames only
		if (head->d_gather)
			(void)hipFree(head->d_gather);
		head->d_gather = nullptr;
		head->gather_floats = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&head->d_gather), gather_total * sizeof(float)));
		head->gather_floats = gather_total;
	}
	HIP_TRY(hipEventRecord(head->ev_begin, stream)); // the peers start once the caller's stream has reached this call
	// (a stream under HIP-graph capture may only wait on events recorded inside the capture: ev_gathered, which orders this render
	// against the PREVIOUS one, is neither waited on nor recorded then -- replays of one graph are ordered by their launch stream)
	hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
	const bool capturing = hipStreamIsCapturing(stream, &capture) == hipSuccess && capture == hipStreamCaptureStatusActive;

	// ---- every member renders its shard ----
	for (uint32_t m = 0; m < n; ++m) {
		rt_scene *mem = members[m];
		if (n_floats[m] == 0)
			continue;
		HIP_TRY(hipSetDevice(mem->device));
		if (n_floats[m] > mem->shard_floats) {
			if (mem->d_shard)
				(void)hipFree(mem->d_shard);
			mem->d_shard = nullptr;
			mem->shard_floats = 0;
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&mem->d_shard), n_floats[m] * sizeof(float)));
			mem->shard_floats = n_floats[m];
		}
		hipStream_t ms = m == 0 ? stream : mem->stream;
		if (m > 0)
			HIP_TRY(hipStreamWaitEvent(ms, head->ev_begin, 0));
		// ... and once the PREVIOUS render's gather has read this member's shard (the head's own shard is read by the scatter on
		// the previous caller stream): a caller that alternates streams between frames must not overwrite a shard in flight
		if (head->gathered_once && !capturing)
			HIP_TRY(hipStreamWaitEvent(ms, head->ev_gathered, 0));
		om.shard_index = m;
		mem->member_call = true;
		const int rc = rt_render_device(mem, camera, &om, mem->d_shard, reinterpret_cast<uint64_t *>(mem->d_member_rays), ms);
		mem->member_call = false;
		if (rc != RT_OK)
			return rc;
		if (m > 0)
			HIP_TRY(hipEventRecord(mem->ev_shard, ms));
	}

	// ---- gather the peers' shards (and ray counters) into the head's HBM: the transport was fixed when the scene was created ----
	HIP_TRY(hipSetDevice(head->device));
	if (head->gather_mode == RT_GATHER_RCCL) {
		void **comms = static_cast<void **>(head->nccl_comms);
		if (rccl().GroupStart() != 0)
			return fail(RT_ERR_HIP, "ncclGroupStart failed");
		int bad = 0;
		for (uint32_t m = 1; m < n; ++m) {
			if (n_floats[m] == 0)
				continue;
			bad |= rccl().Send(members[m]->d_shard, n_floats[m], kNcclFloat32, 0, comms[m], members[m]->stream);
			bad |= rccl().Recv(head->d_gather + offset[m], n_floats[m], kNcclFloat32, (int)m, comms[0], stream);
		}
		if (rccl().GroupEnd() != 0 || bad != 0)
			return fail(RT_ERR_HIP, "RCCL shard gather failed");
		HIP_TRY(hipSetDevice(head->device));
	} else {
		for (uint32_t m = 1; m < n; ++m) {
			if (n_floats[m] == 0)
				continue;
			HIP_TRY(hipStreamWaitEvent(stream, members[m]->ev_shard, 0));
			if (members[m]->device == head->device)
				HIP_TRY(hipMemcpyAsync(head->d_gather + offset[m], members[m]->d_shard, n_floats[m] * sizeof(float), hipMemcpyDeviceToDevice, stream));
			else
				HIP_TRY(hipMemcpyPeerAsync(head->d_gather + offset[m], head->device, members[m]->d_shard, members[m]->device, n_floats[m] * sizeof(float), stream));
		}
	}
	// ---- scatter the shards into the frame (every pixel belongs to exactly one shard: nothing to clear) ----
	for (uint32_t m = 0; m < n; ++m) {
		if (n_floats[m] == 0)
			continue;
		om.shard_index = m;
		ShardGeometry g;
		int rc = shard_geometry(&om, g);
		if (rc != RT_OK)
			return rc;
		DevRenderParams P;
		std::memset(&P, 0, sizeof P);
		P.width = (uint32_t)o->width;
		P.height = (uint32_t)o->height;
		P.shard_index = m;
		P.shard_count = n;
		P.tile_w = g.tile_w;
		P.tile_h = g.tile_h;
		P.tiles_x = g.tiles_x;
		P.tiles_y = g.tiles_y;
		P.n_work = (uint32_t)g.n_work;
		HIP_TRY(launch_scatter_shard(stream, P, m == 0 ? head->d_shard : head->d_gather + offset[m], d_out_rgb));
	}
	if (d_rays_shot) { // SamplerProgress.rays_shot of the whole job
		for (uint32_t m = 0; m < n; ++m) {
			if (n_floats[m] == 0) {
				HIP_TRY(hipMemsetAsync(head->d_gather_rays + m, 0, sizeof(unsigned long long), stream));
			} else if (members[m]->device == head->device) {
				if (m > 0 && head->gather_mode == RT_GATHER_RCCL)
					HIP_TRY(hipStreamWaitEvent(stream, members[m]->ev_shard, 0));
				HIP_TRY(hipMemcpyAsync(head->d_gather_rays + m, members[m]->d_member_rays, sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream));
			} else {
				if (head->gather_mode == RT_GATHER_RCCL)
					HIP_TRY(hipStreamWaitEvent(stream, members[m]->ev_shard, 0));
				HIP_TRY(hipMemcpyPeerAsync(head->d_gather_rays + m, head->device, members[m]->d_member_rays, members[m]->device, sizeof(unsigned long long), stream));
			}
		}
		HIP_TRY(launch_sum_u64(stream, head->d_gather_rays, n, reinterpret_cast<unsigned long long *>(d_rays_shot)));
	}
	// every member's shard and counter have been read once the caller's stream gets here
	if (!capturing) {
		HIP_TRY(hipEventRecord(head->ev_gathered, stream));
		head->gathered_once = true;
	}
	return RT_OK;
}

int rt_render_device(rt_scene *s, const rt_camera *camera, const rt_render_opts *o, float *d_out_rgb, uint64_t *d_rays_shot,
                     void *hip_stream)
{
	if (!s || !camera || !o || !d_out_rgb)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (!s->peers.empty() && !s->member_call) {
		if (s->device == RT_DEVICE_NONE)
			return fail(RT_ERR_NO_DEVICE, "host-only scene");
		return render_device_multi(s, camera, o, d_out_rgb, d_rays_shot, static_cast<hipStream_t>(hip_stream));
	}
	ShardGeometry g;
	int rc = shard_geometry(o, g);
	if (rc != RT_OK)
		return rc;
	if (s->device == RT_DEVICE_NONE)
		return fail(RT_ERR_NO_DEVICE, "host-only scene (RT_DEVICE_NONE): rendering needs a GPU, there is no CPU fallback");
	if (o->render_method != RT_METHOD_NAIVE && o->render_method != RT_METHOD_MIS)
		return fail(RT_ERR_INVALID_ARGUMENT, "unknown render method");
	if (o->samples_per_pixel == 0 || o->samples_per_pixel >= (1ull << 32))
		return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel must be in [1, 2^32)");
	if (o->output_layout != RT_LAYOUT_FRAME && o->output_layout != RT_LAYOUT_SHARD)
		return fail(RT_ERR_INVALID_ARGUMENT, "unknown output layout");
	HIP_TRY(hipSetDevice(s->device));
	hipStream_t stream = static_cast<hipStream_t>(hip_stream);
	if (g.n_work == 0) { // this shard owns no tile (more shards than tiles)
		HIP_TRY(launch_reset(stream, s->d_work_counter, reinterpret_cast<unsigned long long *>(d_rays_shot), d_out_rgb,
		                     o->output_layout == RT_LAYOUT_FRAME ? (size_t)(o->width * o->height * 3) : 0));
		s->timed = false;
		return RT_OK;
	}

	DevRenderParams P;
	std::memset(&P, 0, sizeof P);
	P.width = (uint32_t)o->width;
	P.height = (uint32_t)o->height;
	P.spp = (uint32_t)o->samples_per_pixel;
	P.sample_begin_lo = (uint32_t)o->sample_begin;
	P.sample_begin_hi = (uint32_t)(o->sample_begin >> 32);
	P.seed_lo = (uint32_t)o->seed;
	P.seed_hi = (uint32_t)(o->seed >> 32);
	P.max_depth = o->max_depth;
	P.rr_threshold = o->rr_threshold;
	P.shard_index = o->shard_index;
	P.shard_count = o->shard_count;
	P.tile_w = g.tile_w;
	P.tile_h = g.tile_h;
	P.tiles_x = g.tiles_x;
	P.tiles_y = g.tiles_y;
	P.n_work = (uint32_t)g.n_work;
	// sample_split 0 = automatic, on one device as on several (auto_sample_split: >= 64 work items per resident lane)
	uint32_t split = o->sample_split;
	if (split == 0u)
		split = auto_sample_split(s->n_cus, o->width * o->height, o->samples_per_pixel, o->shard_count);
	{ // u = (jitter + x) / (W - 1), v = (jitter + y) / (H - 1) by verified reciprocals (cached per divisor: rt_build.cpp)
		float rw = 0.0f, rh = 0.0f;
		const bool sized = o->width <= (1u << 24) && o->height <= (1u << 24);
		const bool ok_w = sized && verified_reciprocal((float)(o->width - 1), &rw), ok_h = sized && verified_reciprocal((float)(o->height - 1), &rh);
		const bool ok = ok_w && ok_h;
		P.w1h1_ok = ok ? 1u : 0u;
		P.inv_w1 = rw;
		P.inv_h1 = rh;
	}
	P.tile_log2_w = 0xFFFFFFFFu;
	if (g.tile_w * g.tile_h == 64u && (g.tile_w & (g.tile_w - 1u)) == 0u && o->width < 65536u && o->height < 65536u &&
	    (split & (split - 1u)) == 0u && split <= 64u)
		for (uint32_t lw = 0; lw < 7u; ++lw)
			for (uint32_t ls = 0; ls < 7u; ++ls)
				if ((1u << lw) == g.tile_w && (1u << ls) == split)
					P.tile_log2_w = lw | (ls << 8);
	if (split > o->samples_per_pixel)
		return fail(RT_ERR_INVALID_ARGUMENT, "sample_split larger than samples_per_pixel");
	if (g.n_work * split >= (1ull << 32))
		return fail(RT_ERR_UNSUPPORTED, "pixels x sample_split exceeds 2^32 work items");
	P.sample_split = split;
	P.n_items = (uint32_t)(g.n_work * split);
	float *render_target = d_out_rgb;
	if (split > 1u) { // chunk means land in a scratch buffer; combine_chunks_kernel folds them into d_out_rgb
		const size_t need = (size_t)g.n_work * split * 3;
		if (need > s->partial_floats) { // grows on first use only (not capturable into a graph on that call)
			if (s->d_partial)
				(void)hipFree(s->d_partial);
			s->d_partial = nullptr;
			s->partial_floats = 0;
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_partial), need * sizeof(float)));
			s->partial_floats = need;
		}
		render_target = s->d_partial;
	}
	P.shard_layout = o->output_layout == RT_LAYOUT_SHARD ? 1 : 0;

	// traversal: exhaustive (the reference's own amount of work) for tiny trees where pruning cannot
	// pay, t-pruned otherwise; both select the same winner (rt_intersect.h)
	// and how finely the wave votes: measured crossovers on random sphere scenes (tests/probes/gpu_crossover_probe.py)
	bool prune = s->traversal_mode == -1 ? s->dev.n_prims > kPruneAbove : s->traversal_mode == 1;
	// the fine schedule walks the wide tree: scenes without one (non-finite bounds, a single leaf) stay coarse
	const bool fine = s->dev.nodes4 != nullptr &&
	                  (s->schedule_mode == -1 ? (prune && s->dev.n_prims > (s->dev.has_triangles ? kFineAboveTriangles : kFineAboveSpheres))
	                                          : s->schedule_mode == 1);
	if (fine)
		prune = true;
	P.prune = prune ? 1 : 0;

	const bool samplable = (s->dev.sky.res_x | s->dev.sky.res_y) != 0u;
	const size_t sky_bytes = samplable ? ((size_t)s->dev.sky.res_y * (s->dev.sky.res_x + 1u) + s->dev.sky.res_y + 1u) * 4 +
	                                         (size_t)(s->dev.sky.res_y + 1u) * s->dev.sky.guide_k : 0;
	// Sky CDF tables in LDS (next to the traversal stacks) or left in global memory: LDS only while
	// it does not cost resident workgroups.  Tiny trees: 41 KB tables + 1-2 KB stacks still fit 3
	// workgroups per CU, the register limit.  Deep trees: the stacks alone are tens of KB and the
	// sky is a small share of the work, so residency (latency hiding for the node fetches) wins.
	// Tiny scenes under the coarse schedule: the whole scene rides in LDS too.
	const bool scene_lds = !fine && s->dev.blob_bytes != 0u && s->scene_lds_allowed;
	P.scene_in_lds = scene_lds ? 1u : 0u;
	bool sky_lds = false;
	// RT_TUNE_EXCHANGE (rt_render.hip, XCHG).  Fine schedule: 512-thread workgroups whose waves trade whole lane states
	// through two record pools behind the stacks; coarse MIS kernels: decided below, once the occupancy is known
	// (feature set of THIS launch: 3 = FeatPair, built for the exhaustive coarse kernels without the exchange only)
	const int feature_set =
	    (s->feature_set == 0 && s->pair_tree && !s->feature_set_forced && !prune && !fine && s->exchange_mode != 1) ? 3 : s->feature_set;
	bool xchg = s->exchange_mode == 1 && render_exchange_available(o->render_method, prune, fine, feature_set);
	const bool xchg_fine = xchg && fine;
	uint32_t block_threads = render_block_threads(feature_set, fine, xchg_fine);
	const size_t fine_pool_bytes = xchg_fine ? render_exchange_fine_lds_bytes(block_threads / 64u) : 0;
	// Traversal stacks: one LDS column per lane.  The worst case of a deep tree (three pending siblings per level of
	// the wide tree) is far above what walks reach, and LDS sized for it would cost resident waves; under the fine
	// schedule the LDS part is capped at the share a workgroup gets at the occupancy its register budget allows, the
	// rest of the worst case lives in a global overflow area that is touched only if a walk really gets that deep.
	// The coarse kernels keep a walk's WHOLE worst-case stack in LDS.  The wide tree's worst case (three pending siblings
	// per level) is about 1.5 x the two-child tree's, and only pruned walks of regular rays descend it: an exhaustive
	// launch needs the two-child depth only, and a pruned coarse launch whose wide worst case does not fit the LDS of a CU
	// (a deep, skewed tree) walks the two-child tree for every ray instead of failing.
	DevScene dev = s->dev; // what this launch sees
	const bool walks_wide = prune && dev.nodes4 != nullptr && dev.narrow_only == 0u;
	uint32_t stack_need = (fine || walks_wide) ? dev.stack_depth : s->stack_depth_narrow;
	if (!fine && walks_wide &&
	    render_lds_bytes(dev, false, scene_lds, render_block_threads(feature_set, false, false) / 64u, stack_need) > s->max_lds) {
		dev.narrow_only = 1u;
		stack_need = s->stack_depth_narrow;
	}
	uint32_t stack_cap = stack_need;
	if (fine) {
		const uint32_t blocks_wanted = std::max(1u, render_waves_per_simd(feature_set, true) * 256u / block_threads);
		const size_t share = s->max_lds / blocks_wanted;
		const uint32_t fit = (uint32_t)((share > fine_pool_bytes ? share - fine_pool_bytes : 0) / ((block_threads / 64u) * 64u * 4u));
		stack_cap = std::min(stack_cap, std::max(8u, fit));
	}
	if (fine && s->stack_cap_override != 0u) // (coarse kernels keep the whole stack in LDS and walk without capacity checks)
		stack_cap = std::min(stack_need, s->stack_cap_override);
	P.stack_cap = stack_cap;
	P.stack_ovf_depth = stack_need - stack_cap;
	// What one workgroup size gives: resident workgroups per CU without and -- where that does not cost any -- with the sky tables in LDS
	const bool sky_lds_possible = samplable && o->render_method == RT_METHOD_MIS && sky_bytes <= 96 * 1024;
	struct Sizing { uint32_t block; int blocks_per_cu; bool sky; size_t lds; };
	auto size_launch = [&](uint32_t block, Sizing &z) -> hipError_t {
		z.block = block;
		z.sky = false;
		z.blocks_per_cu = 0;
		z.lds = render_lds_bytes(dev, false, scene_lds, block / 64u, stack_cap) + fine_pool_bytes;
		if (z.lds > s->max_lds)
			return hipSuccess; // (does not fit: blocks_per_cu stays 0)
		hipError_t e = render_occupancy(o->render_method, prune, fine, false, feature_set, z.lds, &z.blocks_per_cu, xchg_fine, block);
		if (e != hipSuccess)
			return e;
		if (sky_lds_possible) {
			const size_t lds_with = render_lds_bytes(dev, true, scene_lds, block / 64u, stack_cap) + fine_pool_bytes;
			int blocks_with = 0;
			if (lds_with <= s->max_lds &&
			    render_occupancy(o->render_method, prune, fine, true, feature_set, lds_with, &blocks_with, xchg_fine, block) == hipSuccess &&
			    blocks_with >= z.blocks_per_cu && blocks_with >= 1) {
				z.sky = true;
				z.lds = lds_with;
				z.blocks_per_cu = blocks_with;
			}
		}
		return hipSuccess;
	};
	Sizing Z;
	HIP_TRY(size_launch(block_threads, Z));
	if (Z.lds > s->max_lds)
		return fail(RT_ERR_UNSUPPORTED, "traversal stacks exceed the LDS of one CU");
	// Coarse spheres-only kernels (FeatPair among them) are issue-bound and use 66 - 100 VGPRs: every further wave per SIMD the
	// registers allow is worth about 2 % (rt_render.hip RT_PAIR_WAVES).  Three 512-thread workgroups do not fit the LDS with the
	// sky tables, two of 768 threads do: try the larger workgroup too, keep what puts most waves on a CU, at equal waves what keeps
	// the sky tables in LDS, at equal both the smaller workgroup.  Only multiples of 256 threads: a workgroup whose waves do not
	// divide evenly over the four SIMDs is reported as resident twice by the occupancy query, but the second one is not placed
	// once the fuller SIMDs are out of registers (640 x 2 at 93 VGPRs: 18.8 -> 23.3 ms, 896 x 2: no gain; profiles/r04v_small_ab.log).
	if (!fine && !xchg) {
		for (uint32_t cand = block_threads + 256u; cand <= render_max_block_threads(feature_set, fine, false); cand += 256u) {
			Sizing C;
			if (size_launch(cand, C) != hipSuccess || C.blocks_per_cu < 1)
				continue;
			const uint32_t waves_c = (uint32_t)C.blocks_per_cu * C.block, waves_z = (uint32_t)Z.blocks_per_cu * Z.block;
			if (waves_c > waves_z || (waves_c == waves_z && C.sky && !Z.sky))
				Z = C;
		}
		block_threads = Z.block;
	}
	size_t lds_bytes = Z.lds;
	int blocks_per_cu = Z.blocks_per_cu;
	sky_lds = Z.sky;
	P.sky_in_lds = sky_lds ? 1u : 0u;
	if (blocks_per_cu < 1)
		return fail(RT_ERR_HIP, "render kernel does not fit on a CU");
	// RT_TUNE_EXCHANGE (asked for by name, off by default): the workgroup's pool of parked path states sits behind the stacks
	// (rt_render.hip, XCHG) and gets the LDS that is left at the occupancy the EXCHANGE kernel reaches by its registers -- it never
	// costs the sky tables their place; it can cost a resident workgroup where that kernel needs more registers than the plain one
	// (full feature set: 141 against 125 VGPRs)
	P.xchg_slots = 0;
	if (xchg && !fine) {
		int blocks_plain = 0;
		if (render_occupancy(o->render_method, prune, fine, sky_lds, feature_set, lds_bytes, &blocks_plain, true) != hipSuccess || blocks_plain < 1) {
			xchg = false;
		} else {
			const int target = std::min(blocks_per_cu, blocks_plain);
			const size_t share = s->max_lds / (size_t)target; // max_lds is the LDS of one CU
			uint32_t slots = render_exchange_max_slots();
			while (slots >= 16u && lds_bytes + render_exchange_lds_bytes(slots) > share)
				slots -= 4u;
			int blocks_x = 0;
			if (slots >= 16u &&
			    render_occupancy(o->render_method, prune, fine, sky_lds, feature_set, lds_bytes + render_exchange_lds_bytes(slots), &blocks_x, true) == hipSuccess &&
			    blocks_x >= target) {
				P.xchg_slots = slots;
				lds_bytes += render_exchange_lds_bytes(slots);
				blocks_per_cu = target;
			} else {
				xchg = false;
			}
		}
	}
	uint64_t n_blocks = (uint64_t)s->n_cus * (uint64_t)blocks_per_cu;
	const uint64_t blocks_needed = ((uint64_t)P.n_items + block_threads - 1) / block_threads;
	if (n_blocks > blocks_needed)
		n_blocks = blocks_needed ? blocks_needed : 1;

	{ // counters, and the parts of the output no lane will write (other shards' pixels, edge-tile padding)
		size_t zero_floats = 0;
		if (o->output_layout == RT_LAYOUT_FRAME && o->shard_count > 1)
			zero_floats = (size_t)(o->width * o->height * 3);
		if (o->output_layout == RT_LAYOUT_SHARD)
			zero_floats = (size_t)(g.n_work * 3);
		HIP_TRY(launch_reset(stream, s->d_work_counter, reinterpret_cast<unsigned long long *>(d_rays_shot), d_out_rgb, zero_floats));
	}

	DevCamera cam;
	std::memcpy(cam.origin, camera->origin, 12);
	std::memcpy(cam.lower_left, camera->lower_left, 12);
	std::memcpy(cam.horizontal, camera->horizontal, 12);
	std::memcpy(cam.vertical, camera->vertical, 12);

	{ // what is about to run, for rt_last_launch_info
		rt_launch_info &L = s->last_launch;
		std::memset(&L, 0, sizeof L);
		L.method = o->render_method;
		L.pruned = prune ? 1 : 0;
		L.fine = fine ? 1 : 0;
		L.sky_in_lds = sky_lds ? 1 : 0;
		L.scene_in_lds = scene_lds ? 1 : 0;
		L.feature_set = feature_set;
		L.block_threads = block_threads;
		L.n_blocks = (uint32_t)n_blocks;
		L.blocks_per_cu = (uint32_t)blocks_per_cu;
		L.waves_per_simd = (uint32_t)blocks_per_cu * L.block_threads / 256u;
		L.lds_bytes = (uint32_t)lds_bytes;
		L.n_cus = (uint32_t)s->n_cus;
		L.sample_split = split;
		L.n_items = P.n_items;
		static const char *const feat_names[4] = {"rt::Feat<false, false, false, false>", "rt::Feat<true, true, false, false>",
		                                          "rt::Feat<true, true, true, true>", "rt::FeatPair"};
		// pick_render (rt_render.hip) folds these: naive never stages the sky, fine implies pruned
		std::snprintf(L.kernel, sizeof L.kernel, "rt::render_kernel<%d, %s, %s, %s, %s%s>", (int)o->render_method, prune ? "true" : "false",
		              fine ? "true" : "false", (sky_lds && o->render_method == RT_METHOD_MIS) ? "true" : "false", feat_names[feature_set],
		              xchg ? ", true" : ", false"); // the name rocprofv3 prints
	}
	if (P.stack_ovf_depth != 0u) { // grown on first use only (like the sample_split scratch: not capturable on that call)
		const size_t need = (size_t)n_blocks * block_threads * P.stack_ovf_depth;
		if (need > s->stack_ovf_words) {
			if (s->d_stack_ovf)
				(void)hipFree(s->d_stack_ovf);
			s->d_stack_ovf = nullptr;
			s->stack_ovf_words = 0;
			HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_stack_ovf), need * sizeof(uint32_t)));
			s->stack_ovf_words = need;
		}
	}
	HIP_TRY(hipEventRecord(s->ev_start, stream));
	HIP_TRY(launch_render(o->render_method, prune, fine, sky_lds, feature_set, (uint32_t)n_blocks, lds_bytes, stream, dev, cam, P, render_target,
	                      reinterpret_cast<unsigned long long *>(d_rays_shot), s->d_work_counter, s->d_stack_ovf, xchg,
	                      feature_set == 3 ? &s->pair : nullptr, block_threads));
	HIP_TRY(hipEventRecord(s->ev_stop, stream));
	if (split > 1u)
		HIP_TRY(launch_combine(stream, P, s->d_partial, d_out_rgb));
	s->timed = true;
	s->n_launches = 1;
	return RT_OK;
}

int rt_render(rt_scene *s, const rt_camera *camera, const rt_render_opts *o, float *out_rgb, uint64_t *rays_shot)
{
	if (!s || !camera || !o)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (s->device == RT_DEVICE_NONE)
		return fail(RT_ERR_NO_DEVICE, "host-only scene (RT_DEVICE_NONE): this call needs a GPU, there is no CPU fallback");
	uint64_t n_floats = 0;
	int rc = rt_render_output_floats(o, &n_floats);
	if (rc != RT_OK)
		return rc;
	if (n_floats == 0) { // a shard that owns no tile: nothing to render, nothing to write
		if (rays_shot)
			*rays_shot = 0;
		return RT_OK;
	}
	if (!out_rgb)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	rc = ensure_frame_buffers(s, n_floats, false); // scene-owned device frame: no allocation per call
	if (rc != RT_OK)
		return rc;
	float *d_out = s->d_prog[0];
	rc = rt_render_device(s, camera, o, d_out, reinterpret_cast<uint64_t *>(s->d_rays), s->stream);
	if (rc == RT_OK) {
		hipError_t e = hipMemcpyAsync(out_rgb, d_out, n_floats * sizeof(float), hipMemcpyDeviceToHost, s->stream);
		if (e == hipSuccess && rays_shot)
			e = hipMemcpyAsync(rays_shot, s->d_rays, sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream);
		if (e == hipSuccess)
			e = hipStreamSynchronize(s->stream);
		if (e != hipSuccess)
			rc = hip_fail(e, "render");
	}
	return rc;
}

int rt_output_rgb8_device(rt_scene *s, const float *d_rgb, uint64_t n_values, float gamma, uint8_t *d_out, void *hip_stream)
{
	if (!s || !d_rgb || !d_out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (s->device == RT_DEVICE_NONE)
		return fail(RT_ERR_NO_DEVICE, "host-only scene (RT_DEVICE_NONE): this call needs a GPU (rt_output_rgb8 is the host-side conversion)");
	if (n_values == 0)
		return RT_OK;
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(launch_quantise(static_cast<hipStream_t>(hip_stream), d_rgb, (size_t)n_values, 1.0f / gamma, d_out));
	return RT_OK;
}

int rt_render_rgb8(rt_scene *s, const rt_camera *camera, const rt_render_opts *o, float gamma, uint8_t *out_rgb8, uint64_t *rays_shot)
{
	if (!s || !camera || !o)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (s->device == RT_DEVICE_NONE)
		return fail(RT_ERR_NO_DEVICE, "host-only scene (RT_DEVICE_NONE): this call needs a GPU, there is no CPU fallback");
	uint64_t n_values = 0;
	int rc = rt_render_output_floats(o, &n_values);
	if (rc != RT_OK)
		return rc;
	if (n_values == 0) {
		if (rays_shot)
			*rays_shot = 0;
		return RT_OK;
	}
	if (!out_rgb8)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	HIP_TRY(hipSetDevice(s->device));
	rc = ensure_frame_buffers(s, n_values, false);
	if (rc != RT_OK)
		return rc;
	if (n_values > s->d_rgb8_bytes) { // scene-owned byte frame, grown on first use
		if (s->d_rgb8)
			(void)hipFree(s->d_rgb8);
		s->d_rgb8 = nullptr;
		s->d_rgb8_bytes = 0;
		HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_rgb8), n_values));
		s->d_rgb8_bytes = n_values;
	}
	float *d_frame = s->d_prog[0];
	rc = rt_render_device(s, camera, o, d_frame, reinterpret_cast<uint64_t *>(s->d_rays), s->stream);
	if (rc != RT_OK)
		return rc;
	// the output stage runs where the frame is: the float frame never crosses PCIe, one byte per value does
	hipError_t e = launch_quantise(s->stream, d_frame, (size_t)n_values, 1.0f / gamma, s->d_rgb8);
	if (e == hipSuccess)
		e = hipMemcpyAsync(out_rgb8, s->d_rgb8, n_values, hipMemcpyDeviceToHost, s->stream);
	if (e == hipSuccess && rays_shot)
		e = hipMemcpyAsync(rays_shot, s->d_rays, sizeof(uint64_t), hipMemcpyDeviceToHost, s->stream);
	if (e == hipSuccess)
		e = hipStreamSynchronize(s->stream);
	if (e != hipSuccess)
		return hip_fail(e, "render_rgb8");
	return RT_OK;
}

int rt_sample_image(rt_scene *s, const rt_camera *camera, const rt_render_opts *o, uint64_t batch, rt_presentation_update update, void *data)
{
	if (!s || !camera || !o)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (s->device == RT_DEVICE_NONE)
		return fail(RT_ERR_NO_DEVICE, "host-only scene (RT_DEVICE_NONE): this call needs a GPU, there is no CPU fallback");
	uint64_t n_floats = 0;
	int rc = rt_render_output_floats(o, &n_floats);
	if (rc != RT_OK)
		return rc;
	const uint64_t spp = o->samples_per_pixel;
	if (spp == 0 || spp >= (1ull << 32))
		return fail(RT_ERR_INVALID_ARGUMENT, "samples_per_pixel must be in [1, 2^32)");
	if (batch == 0 || batch > spp)
		batch = spp;
	HIP_TRY(hipSetDevice(s->device));
	rc = ensure_frame_buffers(s, n_floats, true);
	if (rc != RT_OK)
		return rc;

	const uint64_t n_batches = (spp + batch - 1) / batch;
	uint64_t delivered = 0;
	// deliver batch j (its copy was enqueued on copy_stream): wait for the copy, run the callback
	auto deliver = [&](uint64_t j, bool &cancel) -> int {
		const int b = (int)(j & 1);
		HIP_TRY(hipEventSynchronize(s->ev_copy[b]));
		const uint64_t nb = std::min<uint64_t>(batch, spp - j * batch);
		delivered += nb;
		cancel = false;
		if (update) {
			rt_sampler_progress p;
			p.samples_completed = nb;
			p.rays_shot = s->h_prog_rays[b];
			p.current_image = s->h_prog[b];
			p.n_floats = n_floats;
			cancel = update(data, &p, delivered) != 0;
		}
		return RT_OK;
	};
	for (uint64_t j = 0; j < n_batches; ++j) {
		const int b = (int)(j & 1);
		rt_render_opts oj = *o;
		oj.samples_per_pixel = std::min<uint64_t>(batch, spp - j * batch);
		oj.sample_begin = o->sample_begin + j * batch;
		if (oj.sample_split > oj.samples_per_pixel)
			oj.sample_split = (uint32_t)oj.samples_per_pixel;
		rc = rt_render_device(s, camera, &oj, s->d_prog[b], reinterpret_cast<uint64_t *>(s->d_prog_rays + b), s->stream);
		if (rc != RT_OK)
			break;
		HIP_TRY(hipEventRecord(s->ev_batch[b], s->stream));
		HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_batch[b], 0));
		if (n_floats)
			HIP_TRY(hipMemcpyAsync(s->h_prog[b], s->d_prog[b], n_floats * sizeof(float), hipMemcpyDeviceToHost, s->copy_stream));
		HIP_TRY(hipMemcpyAsync(s->h_prog_rays + b, s->d_prog_rays + b, sizeof(unsigned long long), hipMemcpyDeviceToHost, s->copy_stream));
		HIP_TRY(hipEventRecord(s->ev_copy[b], s->copy_stream));
		if (j > 0) { // the previous batch goes to the callback while this one renders (random_sampler.rs:82-90)
			bool cancel = false;
			rc = deliver(j - 1, cancel);
			if (rc != RT_OK)
				break;
			if (cancel) { // `return` inside the loop: the batch in flight is dropped
				HIP_TRY(hipStreamSynchronize(s->stream));
				HIP_TRY(hipStreamSynchronize(s->copy_stream));
				return RT_OK;
			}
		}
	}
	if (rc != RT_OK) {
		(void)hipStreamSynchronize(s->stream);
		(void)hipStreamSynchronize(s->copy_stream);
		return rc;
	}
	bool ignored = false; // random_sampler.rs:92-98: the last image, return value ignored
	return deliver(n_batches - 1, ignored);
}

int rt_last_kernel_ms(rt_scene *s, float *ms, uint32_t *n_launches)
{
	if (!s || !ms)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (!s->timed)
		return fail(RT_ERR_INVALID_ARGUMENT, "no render has been launched on this scene");
	HIP_TRY(hipSetDevice(s->device));
	HIP_TRY(hipEventSynchronize(s->ev_stop));
	HIP_TRY(hipEventElapsedTime(ms, s->ev_start, s->ev_stop));
	for (rt_scene *m : s->peers) { // multi-device scene: the slowest member's kernel
		if (!m->timed)
			continue;
		float mm = 0.0f;
		HIP_TRY(hipSetDevice(m->device));
		HIP_TRY(hipEventSynchronize(m->ev_stop));
		HIP_TRY(hipEventElapsedTime(&mm, m->ev_start, m->ev_stop));
		*ms = std::max(*ms, mm);
	}
	if (!s->peers.empty())
		HIP_TRY(hipSetDevice(s->device));
	if (n_launches)
		*n_launches = s->n_launches;
	return RT_OK;
}

int rt_last_launch_info(const rt_scene *s, rt_launch_info *out)
{
	if (!s || !out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (!s->timed)
		return fail(RT_ERR_INVALID_ARGUMENT, "no render has been launched on this scene");
	*out = s->last_launch;
	return RT_OK;
}

// ---- output stage (host only): crates/output/src/lib.rs:74-113 ----
namespace {

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
	static uint32_t table[256];
	static bool init = false;
	if (!init) {
		for (uint32_t i = 0; i < 256; ++i) {
			uint32_t c = i;
			for (int k = 0; k < 8; ++k)
				c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			table[i] = c;
		}
		init = true;
	}
	for (size_t i = 0; i < n; ++i)
		crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
	return crc;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
	v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void put_le32(std::vector<uint8_t> &v, uint32_t x)
{
	v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 24));
}
void put_str0(std::vector<uint8_t> &v, const char *s)
{
	v.insert(v.end(), s, s + std::strlen(s) + 1);
}
void png_chunk(std::vector<uint8_t> &png, const char type[4], const std::vector<uint8_t> &data)
{
	put_be32(png, (uint32_t)data.size());
	const size_t start = png.size();
	png.insert(png.end(), type, type + 4);
	png.insert(png.end(), data.begin(), data.end());
	put_be32(png, crc32_update(0xFFFFFFFFu, png.data() + start, png.size() - start) ^ 0xFFFFFFFFu);
}

} // namespace

extern "C" {

int rt_output_rgb8(const float *rgb, uint64_t n_values, float gamma, uint8_t *out)
{
	if (!rgb || !out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	const float inv_gamma = 1.0f / gamma;
	for (uint64_t i = 0; i < n_values; ++i)
		out[i] = rt_quantise_u8(rgb[i], inv_gamma); // (val.powf(1.0 / gamma) * 255.999) as u8, powf = the contract's rt_powf
	return RT_OK;
}

int rt_output_save(const char *filename, const float *rgb, uint32_t width, uint32_t height, float gamma)
{
	if (!filename || !rgb || width == 0 || height == 0)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments");
	const std::string name(filename);
	// save_data_to_image splits on '.', demands exactly one, and dispatches on the extension (lib.rs:81-88)
	const size_t dot = name.find('.');
	if (dot == std::string::npos || name.find('.', dot + 1) != std::string::npos)
		return fail(RT_ERR_INVALID_ARGUMENT, "Invalid filename: exactly one '.' expected");
	const std::string ext = name.substr(dot + 1);
	const uint64_t n = (uint64_t)width * height * 3;
	std::vector<uint8_t> file;
	if (ext == "exr") {
		// "gamma is ignored because of exr" (lib.rs:99-106): the float image itself, as an uncompressed
		// scanline OpenEXR file with FLOAT channels B, G, R
		static const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};
		file.assign(magic, magic + 8);
		auto attr = [&](const char *aname, const char *type, const std::vector<uint8_t> &value) {
			put_str0(file, aname);
			put_str0(file, type);
			put_le32(file, (uint32_t)value.size());
			file.insert(file.end(), value.begin(), value.end());
		};
		std::vector<uint8_t> v;
		for (const char *ch : {"B", "G", "R"}) {
			put_str0(v, ch);
			put_le32(v, 2); // FLOAT
			put_le32(v, 0); // pLinear + 3 reserved bytes
			put_le32(v, 1); // xSampling
			put_le32(v, 1); // ySampling
		}
		v.push_back(0);
		attr("channels", "chlist", v);
		attr("compression", "compression", {0});
		v.clear();
		put_le32(v, 0); put_le32(v, 0); put_le32(v, width - 1); put_le32(v, height - 1);
		attr("dataWindow", "box2i", v);
		attr("displayWindow", "box2i", v);
		attr("lineOrder", "lineOrder", {0});
		v.clear();
		put_le32(v, 0x3F800000u);
		attr("pixelAspectRatio", "float", v);
		attr("screenWindowWidth", "float", v);
		v.clear();
		put_le32(v, 0); put_le32(v, 0);
		attr("screenWindowCenter", "v2f", v);
		file.push_back(0);
		const uint64_t row_bytes = (uint64_t)width * 12;
		const uint64_t first = file.size() + (uint64_t)height * 8;
		for (uint32_t y = 0; y < height; ++y) {
			const uint64_t off = first + (uint64_t)y * (8 + row_bytes);
			put_le32(file, (uint32_t)off);
			put_le32(file, (uint32_t)(off >> 32));
		}
		file.reserve(file.size() + (size_t)height * (8 + row_bytes));
		for (uint32_t y = 0; y < height; ++y) {
			put_le32(file, y);
			put_le32(file, (uint32_t)row_bytes);
			for (int c = 2; c >= 0; --c) // B, G, R planes of the scanline
				for (uint32_t x = 0; x < width; ++x) {
					uint32_t bits;
					std::memcpy(&bits, &rgb[((size_t)y * width + x) * 3 + c], 4);
					put_le32(file, bits);
				}
		}
	} else if (ext == "ppm" || ext == "png" || ext == "bmp" || ext == "tiff") {
		std::vector<uint8_t> px(n);
		rt_output_rgb8(rgb, n, gamma, px.data());
		if (ext == "ppm") {
			char header[64];
			const int len = std::snprintf(header, sizeof header, "P6\n%u %u\n255\n", width, height);
			file.assign(header, header + len);
			file.insert(file.end(), px.begin(), px.end());
		} else if (ext == "bmp") {
			// 24-bit BI_RGB, bottom-up rows of B,G,R padded to 4 bytes
			const uint32_t stride = (width * 3 + 3) & ~3u;
			const uint32_t size = 54 + stride * height;
			file.push_back('B'); file.push_back('M');
			put_le32(file, size); put_le32(file, 0); put_le32(file, 54);
			put_le32(file, 40); put_le32(file, width); put_le32(file, height);
			put_le32(file, 1u | (24u << 16)); // planes, bits per pixel
			put_le32(file, 0); put_le32(file, stride * height);
			put_le32(file, 2835); put_le32(file, 2835); put_le32(file, 0); put_le32(file, 0);
			file.resize(size, 0);
			for (uint32_t y = 0; y < height; ++y) {
				uint8_t *row = file.data() + 54 + (size_t)(height - 1 - y) * stride;
				for (uint32_t x = 0; x < width; ++x)
					for (int c = 0; c < 3; ++c)
						row[x * 3 + c] = px[((size_t)y * width + x) * 3 + (2 - c)];
			}
		} else if (ext == "tiff") {
			// little-endian baseline TIFF: one uncompressed RGB strip, then the IFD
			const uint32_t strip = 8, bits_at = strip + (uint32_t)n, ifd_at = (bits_at + 6 + 1) & ~1u;
			file.push_back('I'); file.push_back('I'); file.push_back(42); file.push_back(0);
			put_le32(file, ifd_at);
			file.insert(file.end(), px.begin(), px.end());
			for (int c = 0; c < 3; ++c) { file.push_back(8); file.push_back(0); }
			file.resize(ifd_at, 0);
			struct Tag { uint16_t id, type; uint32_t count, value; };
			const Tag tags[] = {{256, 4, 1, width}, {257, 4, 1, height}, {258, 3, 3, bits_at}, {259, 3, 1, 1}, {262, 3, 1, 2},
			                    {273, 4, 1, strip}, {277, 3, 1, 3}, {278, 4, 1, height}, {279, 4, 1, (uint32_t)n}, {284, 3, 1, 1}};
			const uint16_t n_tags = sizeof tags / sizeof tags[0];
			file.push_back((uint8_t)n_tags); file.push_back(0);
			for (const Tag &t : tags) {
				file.push_back((uint8_t)t.id); file.push_back((uint8_t)(t.id >> 8));
				file.push_back((uint8_t)t.type); file.push_back(0);
				put_le32(file, t.count);
				put_le32(file, t.value); // SHORT values sit in the low half of the little-endian field
			}
			put_le32(file, 0);
		} else {
			static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
			file.assign(sig, sig + 8);
			std::vector<uint8_t> ihdr;
			put_be32(ihdr, width);
			put_be32(ihdr, height);
			const uint8_t rest[5] = {8, 2, 0, 0, 0}; // 8 bits, RGB
			ihdr.insert(ihdr.end(), rest, rest + 5);
			png_chunk(file, "IHDR", ihdr);
			// scanlines with filter byte 0, wrapped in zlib "stored" blocks
			std::vector<uint8_t> raw;
			raw.reserve((size_t)height * (width * 3 + 1));
			for (uint32_t y = 0; y < height; ++y) {
				raw.push_back(0);
				raw.insert(raw.end(), px.begin() + (size_t)y * width * 3, px.begin() + (size_t)(y + 1) * width * 3);
			}
			std::vector<uint8_t> z;
			z.push_back(0x78);
			z.push_back(0x01);
			uint32_t a = 1, b = 0; // adler32
			for (size_t pos = 0; pos < raw.size();) {
				const size_t len = std::min<size_t>(65535, raw.size() - pos);
				z.push_back(pos + len == raw.size() ? 1 : 0);
				z.push_back((uint8_t)(len & 0xFF)); z.push_back((uint8_t)(len >> 8));
				z.push_back((uint8_t)(~len & 0xFF)); z.push_back((uint8_t)((~len >> 8) & 0xFF));
				z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
				for (size_t i = pos; i < pos + len; ++i) {
					a = (a + raw[i]) % 65521u;
					b = (b + a) % 65521u;
				}
				pos += len;
			}
			put_be32(z, (b << 16) | a);
			png_chunk(file, "IDAT", z);
			png_chunk(file, "IEND", {});
		}
	} else {
		// the reference also hands jpg/jpeg to the image crate; no JPEG encoder here
		return fail(RT_ERR_UNSUPPORTED, "Unable to save file: (unknown or unsupported filetype ." + ext + ")");
	}
	FILE *f = std::fopen(filename, "wb");
	if (!f)
		return fail(RT_ERR_INVALID_ARGUMENT, "cannot open output file");
	const size_t written = std::fwrite(file.data(), 1, file.size(), f);
	std::fclose(f);
	if (written != file.size())
		return fail(RT_ERR_INVALID_ARGUMENT, "short write");
	return RT_OK;
}

} // extern "C"

#ifdef RT_STATS
// diagnostic build only (tests/probes/gpu_trace_queue.py): n rays through the traversal-only persistent kernel at `waves`
// waves/SIMD with `cap` stack entries per lane in LDS; returns (t, primitive) per ray, the kernel time and the node steps
extern "C" int rt_debug_trace_queue(rt_scene *s, const rt_ray_desc *rays, uint64_t n, int waves, uint32_t cap, float *out_t, uint32_t *out_prim,
                                    float *ms, unsigned long long *node_steps)
{
	if (!s || !rays || !out_t || !out_prim || n == 0 || n >= (1ull << 31) || s->device == RT_DEVICE_NONE || s->dev.nodes4 == nullptr)
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments (or no wide tree)");
	HIP_TRY(hipSetDevice(s->device));
	cap = std::min(std::max(cap, 1u), s->dev.stack_depth);
	const uint32_t ovf_depth = s->dev.stack_depth - cap;
	const uint32_t n_blocks = (uint32_t)s->n_cus * (uint32_t)waves;
	const size_t lds_bytes = (size_t)4 * cap * 64 * 4;
	void *d_rays = nullptr, *d_out = nullptr, *d_misc = nullptr, *d_ovf = nullptr;
	HIP_TRY(hipMalloc(&d_rays, n * sizeof(rt_ray_desc)));
	HIP_TRY(hipMalloc(&d_out, n * 8));
	HIP_TRY(hipMalloc(&d_misc, 16));
	HIP_TRY(hipMalloc(&d_ovf, std::max<size_t>(16, (size_t)n_blocks * 256 * ovf_depth * 4)));
	HIP_TRY(hipMemcpy(d_rays, rays, n * sizeof(rt_ray_desc), hipMemcpyHostToDevice));
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		HIP_TRY(hipMemset(d_misc, 0, 16));
		HIP_TRY(hipEventRecord(s->ev_start, s->stream));
		HIP_TRY(launch_trace_queue(waves, n_blocks, lds_bytes, s->stream, s->dev, d_rays, (uint32_t)n, d_out, static_cast<uint32_t *>(d_misc),
		                           reinterpret_cast<unsigned long long *>(static_cast<char *>(d_misc) + 8), cap, ovf_depth, static_cast<uint32_t *>(d_ovf)));
		HIP_TRY(hipEventRecord(s->ev_stop, s->stream));
		HIP_TRY(hipEventSynchronize(s->ev_stop));
		float t = 0.0f;
		HIP_TRY(hipEventElapsedTime(&t, s->ev_start, s->ev_stop));
		best = std::min(best, t);
	}
	std::vector<float> tmp(2 * n);
	HIP_TRY(hipMemcpy(tmp.data(), d_out, n * 8, hipMemcpyDeviceToHost));
	for (uint64_t i = 0; i < n; ++i) {
		out_t[i] = tmp[2 * i];
		std::memcpy(&out_prim[i], &tmp[2 * i + 1], 4);
	}
	if (node_steps)
		HIP_TRY(hipMemcpy(node_steps, static_cast<char *>(d_misc) + 8, 8, hipMemcpyDeviceToHost));
	if (ms)
		*ms = best;
	(void)hipFree(d_rays); (void)hipFree(d_out); (void)hipFree(d_misc); (void)hipFree(d_ovf);
	return RT_OK;
}
#endif

// ---- batch hit queries ----
static int check_common(rt_scene *s, const rt_ray_desc *rays, const uint64_t *object_index, uint64_t n, rt_hit_record *out)
{
	if (!s || !rays || !out)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	if (s->device == RT_DEVICE_NONE)
		return fail(RT_ERR_NO_DEVICE, "host-only scene (RT_DEVICE_NONE): this call needs a GPU, there is no CPU fallback");
	if (n == 0)
		return RT_OK;
	if (object_index)
		for (uint64_t i = 0; i < n; ++i)
			if (object_index[i] >= s->dev.n_prims)
				return fail(RT_ERR_INVALID_ARGUMENT, "object index out of range");
	HIP_TRY(hipSetDevice(s->device));
	void *d_rays = nullptr, *d_out = nullptr, *d_idx = nullptr;
	HIP_TRY(hipMalloc(&d_rays, n * sizeof(rt_ray_desc)));
	hipError_t e = hipMalloc(&d_out, n * sizeof(rt_hit_record));
	if (e == hipSuccess && object_index)
		e = hipMalloc(&d_idx, n * sizeof(uint64_t));
	if (e == hipSuccess)
		e = hipMemcpyAsync(d_rays, rays, n * sizeof(rt_ray_desc), hipMemcpyHostToDevice, s->stream);
	if (e == hipSuccess && object_index)
		e = hipMemcpyAsync(d_idx, object_index, n * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream);
	const bool prune = s->traversal_mode == -1 ? s->dev.n_prims > kPruneAbove : s->traversal_mode == 1;
	// the batch kernels keep the whole worst-case stack of four waves in LDS: the wide tree's only where it is walked and
	// fits, the two-child tree's (and the two-child walk for every ray) otherwise -- as rt_render_device does
	DevScene dev = s->dev;
	const bool walks_wide = prune && dev.nodes4 != nullptr && dev.narrow_only == 0u;
	if (!walks_wide || (size_t)4 * dev.stack_depth * 64u * sizeof(uint32_t) > s->max_lds) {
		if (walks_wide)
			dev.narrow_only = 1u;
		dev.stack_depth = s->stack_depth_narrow;
	}
	if (e == hipSuccess)
		e = object_index ? launch_check_hit_index(prune, s->stream, dev, d_rays, d_idx, n, d_out)
		                 : launch_check_hit(prune, s->stream, dev, d_rays, n, d_out);
	if (e == hipSuccess)
		e = hipMemcpyAsync(out, d_out, n * sizeof(rt_hit_record), hipMemcpyDeviceToHost, s->stream);
	if (e == hipSuccess)
		e = hipStreamSynchronize(s->stream);
	(void)hipFree(d_rays);
	(void)hipFree(d_out);
	(void)hipFree(d_idx);
	if (e != hipSuccess)
		return hip_fail(e, "check_hit");
	return RT_OK;
}

int rt_check_hit(rt_scene *s, const rt_ray_desc *rays, uint64_t n_rays, rt_hit_record *out)
{
	return check_common(s, rays, nullptr, n_rays, out);
}
int rt_check_hit_index(rt_scene *s, const rt_ray_desc *rays, const uint64_t *object_index, uint64_t n_rays, rt_hit_record *out)
{
	if (!object_index)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	return check_common(s, rays, object_index, n_rays, out);
}

int rt_selftest_division(float divisor, float *reciprocal, int *exact)
{
	if (!reciprocal || !exact)
		return fail(RT_ERR_INVALID_ARGUMENT, "null argument");
	*exact = verified_reciprocal(divisor, reciprocal) ? 1 : 0;
	return RT_OK;
}

int rt_selftest_lean(int device, uint64_t n_per_thread, uint64_t seed, uint64_t mismatches[RT_SELFTEST_LEAN_CLASSES])
{
	if (!mismatches || n_per_thread == 0 || n_per_thread > (1ull << 20))
		return fail(RT_ERR_INVALID_ARGUMENT, "bad arguments (n_per_thread in [1, 2^20])");
	int n_dev = 0;
	if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
		return fail(RT_ERR_NO_DEVICE, "no such HIP device");
	HIP_TRY(hipSetDevice(device));
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc(reinterpret_cast<void **>(&d), RT_SELFTEST_LEAN_CLASSES * sizeof(unsigned long long)));
	hipError_t e = hipMemset(d, 0, RT_SELFTEST_LEAN_CLASSES * sizeof(unsigned long long));
	if (e == hipSuccess)
		e = launch_selftest_lean(nullptr, 1024u, n_per_thread, seed, d);
	if (e == hipSuccess)
		e = hipMemcpy(mismatches, d, RT_SELFTEST_LEAN_CLASSES * sizeof(unsigned long long), hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess)
		return hip_fail(e, "selftest");
	return RT_OK;
}

} // extern "C"
