#include "rt_build.h"
#include <cstring>
using namespace rt;
extern "C" int dbg_build(const rt_scene_desc *d, rt_bvh_node *out, uint64_t cap, uint64_t *n, uint64_t *order, uint64_t ocap)
{
	HostScene hs; std::string err;
	int rc = build_host_scene(d, hs, err);
	if (rc) { fprintf(stderr, "err %s\n", err.c_str()); return rc; }
	*n = hs.nodes.size();
	for (size_t i = 0; i < hs.nodes.size() && i < cap; ++i) {
		const HostNode &h = hs.nodes[i];
		memcpy(out[i].min, h.min, 12); memcpy(out[i].max, h.max, 12);
		out[i].children[0] = h.child[0]; out[i].children[1] = h.child[1];
		out[i].primitive_offset = h.primitive_offset; out[i].number_primitives = h.number_primitives;
	}
	for (size_t i = 0; i < hs.primitive_order.size() && i < ocap; ++i) order[i] = hs.primitive_order[i];
	return 0;
}
