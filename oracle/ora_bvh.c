/*
 * ora_bvh.c -- CPU oracle: BVH build (SAH / middle / equal-counts), breadth-first candidate
 * collection and the two hit queries.  TEST INFRASTRUCTURE (see ora_internal.h).
 * Restates crates/implementations/src/acceleration/{mod,split}.rs.
 */
#include "ora_internal.h"
#include <stdlib.h>
#include <string.h>

#define NUM_BUCKETS 12  /* split.rs:5 */
#define MAX_IN_NODE 255 /* split.rs:6 */

/* ---- AABB::merge / extend_contains  aabb.rs:59-77 (Option<AABB> = has flag) ---- */
static void aabb_merge(ora_aabb *acc, bool *has, ora_aabb second)
{
	if (*has) {
		acc->min = v3_min_by_component(acc->min, second.min);
		acc->max = v3_max_by_component(acc->max, second.max);
	} else {
		*acc = second;
		*has = true;
	}
}
static void aabb_extend_contains(ora_aabb *acc, bool *has, vec3 point)
{
	if (*has) {
		acc->min = v3_min_by_component(acc->min, point);
		acc->max = v3_max_by_component(acc->max, point);
	} else {
		acc->min = point;
		acc->max = point;
		*has = true;
	}
}
static float aabb_surface_area(const ora_aabb *b) /* aabb.rs:83-86 */
{
	const vec3 e = v3_sub(b->max, b->min);
	return 2.0f * (e.x * e.y + e.x * e.z + e.y * e.z);
}

/* Axis::get_axis_value / get_max_axis  primitives/mod.rs:29-60 */
static inline float axis_value(int axis, vec3 p) { return axis == 0 ? p.x : (axis == 1 ? p.y : p.z); }
static inline int get_max_axis(vec3 v)
{
	if (v.x > v.y && v.x > v.z)
		return 0;
	if (v.y > v.z)
		return 1;
	return 2;
}

/* slice::sort_by(|a,b| a.center[axis].partial_cmp(b.center[axis])) is a STABLE sort; any stable
 * sort yields the same permutation, here a top-down merge sort. */
static void stable_sort_by_axis(ora_prim_info *a, uint64_t n, int axis, ora_prim_info *tmp)
{
	if (n < 2)
		return;
	if (n <= 8) { /* insertion sort (stable) */
		for (uint64_t i = 1; i < n; ++i) {
			const ora_prim_info key = a[i];
			const float kv = axis_value(axis, key.center);
			uint64_t j = i;
			while (j > 0 && axis_value(axis, a[j - 1].center) > kv) {
				a[j] = a[j - 1];
				--j;
			}
			a[j] = key;
		}
		return;
	}
	const uint64_t mid = n / 2;
	stable_sort_by_axis(a, mid, axis, tmp);
	stable_sort_by_axis(a + mid, n - mid, axis, tmp);
	memcpy(tmp, a, mid * sizeof *a);
	uint64_t i = 0, j = mid, k = 0;
	while (i < mid && j < n) {
		/* take from the right run only when strictly smaller: keeps equal keys in order */
		if (axis_value(axis, a[j].center) < axis_value(axis, tmp[i].center))
			a[k++] = a[j++];
		else
			a[k++] = tmp[i++];
	}
	while (i < mid)
		a[k++] = tmp[i++];
}

/* split.rs:201-210 */
static uint64_t split_equal(int axis, ora_prim_info *info, uint64_t len, ora_prim_info *tmp)
{
	stable_sort_by_axis(info, len, axis, tmp);
	return len / 2;
}

/* split.rs:189-199.  `as usize` is a saturating cast (NaN and negatives -> 0). */
static uint64_t calculate_b(int axis, const ora_prim_info *pi, float min, float extent)
{
	const float absolute_value = axis_value(axis, pi->center);
	const float f = (float)NUM_BUCKETS * (absolute_value - min) / extent;
	uint64_t b;
	if (!(f > 0.0f))
		b = 0;
	else if (f >= 1.8446744e19f)
		b = UINT64_MAX;
	else
		b = (uint64_t)f;
	if (b == NUM_BUCKETS)
		b -= 1;
	return b;
}

/* the partition! macro, split.rs:8-32 (a Hoare-style, UNSTABLE partition; restated step by step
 * because the resulting primitive order decides ties between equal-t hits) */
#define ORA_PARTITION(array, len, PRED, mid_out)                                  \
	do {                                                                          \
		uint64_t left_ = 0, right_ = (len)-1;                                     \
		for (;;) {                                                                \
			while (left_ < (len) && PRED(&(array)[left_]))                        \
				left_ += 1;                                                       \
			while (right_ > 0 && !(PRED(&(array)[right_])))                       \
				right_ -= 1;                                                      \
			if (left_ >= right_) {                                                \
				(mid_out) = left_;                                                \
				break;                                                            \
			}                                                                     \
			const ora_prim_info sw_ = (array)[left_];                             \
			(array)[left_] = (array)[right_];                                     \
			(array)[right_] = sw_;                                                \
		}                                                                         \
	} while (0)

/* SplitType::split  split.rs:78-187 */
static uint64_t split(int split_type, const ora_aabb *bounds, const ora_aabb *center_bounds, int axis,
                      ora_prim_info *info, uint64_t len, ora_prim_info *tmp)
{
	if (split_type == RT_SPLIT_MIDDLE) {
		const float point_mid = 0.5f * (axis_value(axis, center_bounds->min) + axis_value(axis, center_bounds->max));
		uint64_t mid_index;
#define PRED_MIDDLE(pi) (axis_value(axis, (pi)->center) < point_mid)
		ORA_PARTITION(info, len, PRED_MIDDLE, mid_index);
#undef PRED_MIDDLE
		if (mid_index == 0 || mid_index == len - 1)
			stable_sort_by_axis(info, len, axis, tmp);
		return mid_index;
	}
	if (split_type == RT_SPLIT_EQUAL_COUNTS)
		return split_equal(axis, info, len, tmp);

	/* SplitType::Sah */
	if (len <= 4)
		return split_equal(axis, info, len, tmp);

	uint32_t bucket_count[NUM_BUCKETS];
	ora_aabb bucket_bounds[NUM_BUCKETS];
	bool bucket_has[NUM_BUCKETS];
	for (int i = 0; i < NUM_BUCKETS; ++i) {
		bucket_count[i] = 0;
		bucket_has[i] = false;
	}

	const float max_val = axis_value(axis, center_bounds->max);
	const float min_val = axis_value(axis, center_bounds->min);
	const float centroid_extent = max_val - min_val;

	for (uint64_t i = 0; i < len; ++i) {
		const uint64_t b = calculate_b(axis, &info[i], min_val, centroid_extent);
		bucket_count[b] += 1;
		ora_aabb pb;
		pb.min = info[i].min;
		pb.max = info[i].max;
		aabb_merge(&bucket_bounds[b], &bucket_has[b], pb);
	}

	float costs[NUM_BUCKETS - 1];
	for (int i = 0; i < NUM_BUCKETS - 1; ++i) {
		ora_aabb bounds_left, bounds_right;
		bool has_left = false, has_right = false;
		uint32_t count_left = 0, count_right = 0;
		for (int j = 0; j < i + 1; ++j) {
			if (bucket_has[j]) {
				aabb_merge(&bounds_left, &has_left, bucket_bounds[j]);
				count_left += bucket_count[j];
			}
		}
		for (int j = i + 1; j < NUM_BUCKETS; ++j) {
			if (bucket_has[j]) {
				aabb_merge(&bounds_right, &has_right, bucket_bounds[j]);
				count_right += bucket_count[j];
			}
		}
		const float left_sa = has_left ? aabb_surface_area(&bounds_left) : 0.0f;
		const float right_sa = has_right ? aabb_surface_area(&bounds_right) : 0.0f;
		costs[i] = 0.125f + ((float)count_left * left_sa + (float)count_right * right_sa) / aabb_surface_area(bounds);
	}

	float min_cost = costs[0];
	uint64_t min_cost_index = 0;
	for (int i = 1; i < NUM_BUCKETS - 1; ++i) {
		if (costs[i] < min_cost) {
			min_cost = costs[i];
			min_cost_index = (uint64_t)i;
		}
	}

	if (len > MAX_IN_NODE || min_cost < (float)len) {
		uint64_t mid_index;
#define PRED_SAH(pi) (calculate_b(axis, (pi), min_val, centroid_extent) <= min_cost_index)
		ORA_PARTITION(info, len, PRED_SAH, mid_index);
#undef PRED_SAH
		return mid_index;
	}
	return 0;
}

static uint64_t push_node(ora_scene *s, ora_aabb bounds, uint64_t offset, uint64_t count)
{
	if (s->n_nodes == s->cap_nodes) {
		s->cap_nodes = s->cap_nodes ? s->cap_nodes * 2 : 64;
		s->nodes = (ora_node *)realloc(s->nodes, s->cap_nodes * sizeof *s->nodes);
	}
	ora_node *n = &s->nodes[s->n_nodes];
	n->bounds = bounds;
	n->has_children = false;
	n->children[0] = n->children[1] = 0;
	n->primitive_offset = offset;
	n->number_primitives = count;
	return s->n_nodes++;
}

/* Bvh::build_bvh  mod.rs:97-160 */
static uint64_t build_bvh(ora_scene *s, uint64_t offset, ora_prim_info *info, uint64_t number_primitives,
                          ora_prim_info *tmp)
{
	ora_aabb bounds;
	bool has_bounds = false;
	for (uint64_t i = 0; i < number_primitives; ++i) {
		ora_aabb pb;
		pb.min = info[i].min;
		pb.max = info[i].max;
		aabb_merge(&bounds, &has_bounds, pb);
	}

	const uint64_t node_index = push_node(s, bounds, offset, number_primitives);
	bool has_children = false;
	uint64_t child0 = 0, child1 = 0;

	if (number_primitives != 1) {
		ora_aabb center_bounds;
		bool has_cb = false;
		for (uint64_t i = 0; i < number_primitives; ++i)
			aabb_extend_contains(&center_bounds, &has_cb, info[i].center);

		const int axis = get_max_axis(v3_sub(center_bounds.max, center_bounds.min));

		if (fabsf(axis_value(axis, center_bounds.min) - axis_value(axis, center_bounds.max)) <
		    100.0f * ORA_F32_EPSILON) {
			/* leaf holding all primitives */
		} else {
			const uint64_t mid = split(s->split_type, &bounds, &center_bounds, axis, info, number_primitives, tmp);
			if (mid != 0) {
				child0 = build_bvh(s, offset, info, mid, tmp);
				child1 = build_bvh(s, offset + mid, info + mid, number_primitives - mid, tmp);
				has_children = true;
			}
		}
	}

	if (has_children) {
		s->nodes[node_index].has_children = true;
		s->nodes[node_index].children[0] = child0;
		s->nodes[node_index].children[1] = child1;
	}
	return node_index;
}

/* utility::sort_by_indices  utility/mod.rs:119-134, restated with its cycle-following swaps */
void ora_sort_by_indices_u64(uint64_t *vec, uint64_t n, uint64_t *indices)
{
	for (uint64_t index = 0; index < n; ++index) {
		if (indices[index] != index) {
			uint64_t current_index = index;
			for (;;) {
				const uint64_t target_index = indices[current_index];
				indices[current_index] = current_index;
				if (indices[target_index] == target_index)
					break;
				const uint64_t t = vec[current_index];
				vec[current_index] = vec[target_index];
				vec[target_index] = t;
				current_index = target_index;
			}
		}
	}
}

/* Bvh::new  mod.rs:58-93 */
int ora_bvh_build(ora_scene *s)
{
	const uint64_t n = s->n_primitives;
	s->nodes = NULL;
	s->n_nodes = s->cap_nodes = 0;
	s->lights = NULL;
	s->n_lights = 0;
	s->primitive_order = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
	if (n == 0)
		return RT_OK; /* the reference would panic on bounds.unwrap(); callers reject empty scenes */

	ora_prim_info *info = (ora_prim_info *)malloc(n * sizeof *info);
	ora_prim_info *tmp = (ora_prim_info *)malloc(n * sizeof *tmp);
	for (uint64_t i = 0; i < n; ++i) { /* PrimitiveInfo::new  mod.rs:29-41 */
		const ora_aabb b = ora_prim_get_aabb(s, &s->primitives[i]);
		info[i].index = i;
		info[i].min = b.min;
		info[i].max = b.max;
		info[i].center = v3_smul(0.5f, v3_add(b.min, b.max));
	}

	build_bvh(s, 0, info, n, tmp);

	/* sort_by_indices(&mut primitives, info.index): slot i <- old primitive info[i].index */
	ora_primitive *sorted = (ora_primitive *)malloc(n * sizeof *sorted);
	for (uint64_t i = 0; i < n; ++i) {
		sorted[i] = s->primitives[info[i].index];
		s->primitive_order[i] = info[i].index;
	}
	free(s->primitives);
	s->primitives = sorted;
	free(info);
	free(tmp);

	/* lights: mod.rs:84-88 */
	uint64_t n_lights = 0;
	for (uint64_t i = 0; i < n; ++i)
		if (ora_mat_is_light(s, s->primitives[i].material))
			n_lights++;
	s->lights = (uint64_t *)malloc((n_lights ? n_lights : 1) * sizeof(uint64_t));
	for (uint64_t i = 0; i < n; ++i)
		if (ora_mat_is_light(s, s->primitives[i].material))
			s->lights[s->n_lights++] = i;
	return RT_OK;
}

/* ---- per-thread scratch ---- */
void ora_ctx_init(ora_ctx *ctx)
{
	memset(ctx, 0, sizeof *ctx);
	ctx->queue_cap = 256;
	ctx->queue = (uint64_t *)malloc(ctx->queue_cap * sizeof(uint64_t));
	ctx->cand_cap = 128;
	ctx->cand_off = (uint64_t *)malloc(ctx->cand_cap * sizeof(uint64_t));
	ctx->cand_len = (uint64_t *)malloc(ctx->cand_cap * sizeof(uint64_t));
}
void ora_ctx_free(ora_ctx *ctx)
{
	free(ctx->queue);
	free(ctx->cand_off);
	free(ctx->cand_len);
}

/* Bvh::get_intersection_candidates  mod.rs:199-224: breadth-first over every node whose AABB the
 * ray hits; returns the (offset,len) of every hit leaf in BFS order.  No t-ordering, no pruning. */
uint64_t ora_bvh_candidates(const ora_scene *s, const ora_ray *ray, ora_ctx *ctx)
{
	uint64_t n_cand = 0;
	uint64_t head = 0, tail = 0;
	if (s->n_nodes == 0)
		return 0;
	ctx->queue[tail++] = 0;
	while (head != tail) {
		const uint64_t index = ctx->queue[head++];
		const ora_node *node = &s->nodes[index];
		ctx->c.node_tests++;
		if (!ora_aabb_does_int(&node->bounds, ray))
			continue;
		if (node->has_children) {
			if (tail + 2 > ctx->queue_cap) {
				ctx->queue_cap *= 2;
				ctx->queue = (uint64_t *)realloc(ctx->queue, ctx->queue_cap * sizeof(uint64_t));
			}
			ctx->queue[tail++] = node->children[0];
			ctx->queue[tail++] = node->children[1];
		} else {
			if (n_cand == ctx->cand_cap) {
				ctx->cand_cap *= 2;
				ctx->cand_off = (uint64_t *)realloc(ctx->cand_off, ctx->cand_cap * sizeof(uint64_t));
				ctx->cand_len = (uint64_t *)realloc(ctx->cand_len, ctx->cand_cap * sizeof(uint64_t));
			}
			ctx->cand_off[n_cand] = node->primitive_offset;
			ctx->cand_len[n_cand] = node->number_primitives;
			n_cand++;
		}
	}
	return n_cand;
}

/* Bvh::check_hit  mod.rs:265-298 */
uint64_t ora_bvh_check_hit(const ora_scene *s, const ora_ray *ray, ora_si *out, ora_ctx *ctx)
{
	ctx->c.rays++;
	const uint64_t n_cand = ora_bvh_candidates(s, ray, ctx);

	bool have_hit = false;
	ora_si best;
	uint64_t best_index = ORA_NO_INDEX;

	for (uint64_t c = 0; c < n_cand; ++c) {
		const uint64_t offset = ctx->cand_off[c], len = ctx->cand_len[c];
		for (uint64_t index = offset; index < offset + len; ++index) {
			ora_si current;
			if (ora_prim_get_int(s, &s->primitives[index], ray, &current, ctx)) {
				if (current.hit.t > 0.0f) {
					if (have_hit) {
						if (current.hit.t < best.hit.t) {
							best = current;
							best_index = index;
						}
						continue;
					}
					best = current;
					best_index = index;
					have_hit = true;
				}
			}
		}
	}
	if (!have_hit) {
		*out = ora_sky_get_si(s);
		return ORA_NO_INDEX;
	}
	ctx->c.closest_hits++;
	*out = best;
	return best_index;
}

/* Bvh::check_hit_index  mod.rs:226-263 */
bool ora_bvh_check_hit_index(const ora_scene *s, const ora_ray *ray, uint64_t index, ora_si *out, ora_ctx *ctx)
{
	ctx->c.rays++;
	const uint64_t n_cand = ora_bvh_candidates(s, ray, ctx);

	ora_si intersection;
	if (!ora_prim_get_int(s, &s->primitives[index], ray, &intersection, ctx))
		return false;
	if (!(intersection.hit.t > 0.0f))
		return false;
	const float light_t = intersection.hit.t;

	for (uint64_t c = 0; c < n_cand; ++c) {
		const uint64_t offset = ctx->cand_off[c], len = ctx->cand_len[c];
		for (uint64_t current_index = offset; current_index < offset + len; ++current_index) {
			if (current_index == index)
				continue;
			ora_si current;
			if (ora_prim_get_int(s, &s->primitives[current_index], ray, &current, ctx)) {
				if (current.hit.t > 0.0f && current.hit.t < light_t)
					return false;
			}
		}
	}
	*out = intersection;
	return true;
}

/* Bvh::get_pdf_from_index  mod.rs:299-318 */
float ora_bvh_get_pdf_from_index(const ora_scene *s, const ora_hit *last_hit, const ora_hit *light_hit,
                                 vec3 sampled_dir, uint64_t index, ora_ctx *ctx)
{
	const bool sky_samplable = ora_sky_can_sample(s);
	const float divisor = (float)(sky_samplable ? s->n_lights + 1 : s->n_lights);
	if (index == ORA_NO_INDEX)
		return ora_sky_pdf(s, sampled_dir, ctx) / divisor;
	return ora_prim_scattering_pdf(s, &s->primitives[index], last_hit->point, sampled_dir, light_hit) / divisor;
}
