"""Host-side binding of librt_hip.so (the HIP back end) -- the Python twin of the Rust
`extern "C"` block in INTEGRATION.md.

`HipScene` stands where the reference's `Bvh` + `Scene` stand (crates/implementations/src/
acceleration/mod.rs:44-93, src/scene.rs:7-42); `RandomSampler.sample_image` keeps the
reference's trait-method shape (samplers/mod.rs:7-20) including the per-pass callback.

There is no CPU fallback: if the shared library is missing, or no HIP device is present,
every call raises.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")  # RT_HIP_LIB: A/B builds
_LIB = None


class RtHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt_hip error {code}: {message}")
        self.code = code


def lib():
    """Load librt_hip.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RtHipError(abi.RT_ERR_NO_DEVICE, f"{LIB_PATH} not built (run `make -C raytracing-rust_amd/csrc`); "
                             "the HIP back end has no CPU fallback")
        # PyTorch-ROCm ships its own HIP/HSA runtime.  A process that uses both (bench.py, the RCCL gather,
        # the HIP-graph test) must have torch's copy loaded BEFORE librt_hip.so pulls in /opt/rocm's: in the
        # reverse order torch later finds "No HIP GPUs".  So if torch is installed, import it first.
        if "torch" not in sys.modules and os.environ.get("RT_HIP_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _LIB = C.CDLL(LIB_PATH)
        _LIB.rt_last_error.restype = C.c_char_p
        _LIB.rt_abi_version.restype = C.c_uint32
    return _LIB


def _check(rc):
    if rc != 0:
        raise RtHipError(rc, lib().rt_last_error().decode())


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f3(v):
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


def rccl_probe():
    """(usable, note): which RCCL a multi-device scene would bind (rt_rccl_probe; touches no GPU)"""
    usable, note = C.c_int(), C.create_string_buffer(512)
    _check(lib().rt_rccl_probe(C.byref(usable), note, C.c_uint64(512)))
    return bool(usable.value), note.value.decode()


def device_count():
    return int(lib().rt_device_count())


def selftest_lean(device=0, n_per_thread=64, seed=1):
    """rt_selftest_lean: mismatch counts per operand class between the kernels' short arithmetic forms
    (csrc/rt_lean.h) and the plain IEEE operators / rt_detmath.h, evaluated on the GPU.  All must be zero."""
    out = (C.c_uint64 * 9)()
    _check(lib().rt_selftest_lean(C.c_int(device), C.c_uint64(n_per_thread), C.c_uint64(seed), out))
    return [int(x) for x in out]


def camera_new(origin, lookat, vup, fov, aspect_ratio, aperture, focus_dist):
    """SimpleCamera::new (camera.rs:20-54)."""
    cam = abi.Camera()
    _check(lib().rt_camera_new(C.byref(cam), _f3(origin), _f3(lookat), _f3(vup), C.c_float(fov),
                               C.c_float(aspect_ratio), C.c_float(aperture), C.c_float(focus_dist)))
    return cam


HIT_DTYPE = np.dtype([("t", "<f4"), ("point", "<f4", 3), ("error", "<f4", 3), ("normal", "<f4", 3), ("uv", "<f4", 2),
                      ("has_uv", "<i4"), ("out", "<i4"), ("material", "<u4"), ("found", "<u4"), ("index", "<u8")])
NODE_DTYPE = np.dtype([("min", "<f4", 3), ("max", "<f4", 3), ("children", "<i8", 2), ("primitive_offset", "<u8"),
                       ("number_primitives", "<u8")])


WIDE_NODE_DTYPE = np.dtype([("origin", "<f4", 3), ("exps", "<u4"), ("qlo", "<u4", 3), ("qhi", "<u4", 3), ("child", "<u4", 4),
                            ("pad", "<u4", 2)])


def _pack_rays(origins, directions):
    o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.asarray(directions, dtype=np.float32).reshape(-1, 3)
    return np.ascontiguousarray(np.concatenate([o, d], axis=1))


class HipScene:
    """Bvh::new(primitives, sky, split_type) + upload to the HBM of `device`
    (device=abi.RT_DEVICE_NONE: the host-side tree only; rendering then raises RT_ERR_NO_DEVICE)."""

    def __init__(self, scene_description, device=0, devices=None):
        """devices=[d0, d1, ...]: ONE scene replicated over several GPUs (rt_scene_create_multi): every render call shards the
        frame's tiles over them and gathers into d0's HBM."""
        self._desc = scene_description.desc()
        self._h = C.c_void_p()
        if devices is not None:
            devices = [int(d) for d in devices]
            self.device = devices[0]
            self.devices = devices
            arr = (C.c_int * len(devices))(*devices)
            _check(lib().rt_scene_create_multi(C.byref(self._desc), arr, C.c_uint32(len(devices)), C.byref(self._h)))
        else:
            self.device = device
            self.devices = [device]
            _check(lib().rt_scene_create(C.byref(self._desc), C.c_int(device), C.byref(self._h)))

    def device_count(self):
        n = C.c_uint32()
        _check(lib().rt_scene_device_count(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- what Bvh::new produced ----
    def counts(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib().rt_scene_counts(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def nodes(self):
        n = self.counts()[0]
        out = np.zeros(max(1, n), dtype=NODE_DTYPE)
        _check(lib().rt_scene_get_nodes(self._h, out.ctypes.data_as(C.POINTER(abi.BvhNode)), C.c_uint64(n)))
        return out[:n]

    def primitive_order(self):
        n = self.counts()[1]
        out = np.zeros(max(1, n), dtype=np.uint64)
        _check(lib().rt_scene_get_primitive_order(self._h, _p(out, C.c_uint64), C.c_uint64(n)))
        return out[:n]

    def lights(self):
        n = self.counts()[2]
        out = np.zeros(max(1, n), dtype=np.uint64)
        _check(lib().rt_scene_get_lights(self._h, _p(out, C.c_uint64), C.c_uint64(n)))
        return out[:n]

    def wide_tree(self):
        """(wide nodes as a structured array, root ref, stack depth, leaf boxes [n_slots, 8]) -- what pruned walks descend"""
        n, root, depth = C.c_uint64(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_scene_wide_info(self._h, C.byref(n), C.byref(root), C.byref(depth)))
        nodes = np.zeros(max(1, n.value), dtype=WIDE_NODE_DTYPE)
        boxes = np.zeros((max(1, self.counts()[1]), 8), dtype=np.float32)
        if n.value:
            _check(lib().rt_scene_get_wide_nodes(self._h, nodes.ctypes.data_as(C.c_void_p), C.c_uint64(n.value)))
            _check(lib().rt_scene_get_leaf_boxes(self._h, _p(boxes, C.c_float), C.c_uint64(boxes.shape[0])))
        return nodes[:n.value], root.value, depth.value, boxes

    def wide_tree_compact(self):
        """(compact wide nodes -- the bytes the kernels fetch --, leaf boxes by leaf index [n_leaves, 8])"""
        n, root, depth = C.c_uint64(), C.c_uint32(), C.c_uint32()
        _check(lib().rt_scene_wide_info(self._h, C.byref(n), C.byref(root), C.byref(depth)))
        nodes = np.zeros(max(1, n.value), dtype=WIDE_NODE_DTYPE)
        n_leaves = C.c_uint64()
        _check(lib().rt_scene_get_leaf_boxes_compact(self._h, None, C.c_uint64(0), C.byref(n_leaves)))
        boxes = np.zeros((max(1, n_leaves.value), 8), dtype=np.float32)
        if n.value:
            _check(lib().rt_scene_get_wide_nodes_compact(self._h, nodes.ctypes.data_as(C.c_void_p), C.c_uint64(n.value)))
            _check(lib().rt_scene_get_leaf_boxes_compact(self._h, _p(boxes, C.c_float), C.c_uint64(boxes.shape[0]), C.byref(n_leaves)))
        return nodes[:n.value], boxes[:n_leaves.value]

    def set_traversal(self, mode):
        """-1 auto, 0 exhaustive (reference amount of work), 1 pruned."""
        _check(lib().rt_scene_set_traversal(self._h, C.c_int(mode)))

    def set_tuning(self, key, value):
        """abi.RT_TUNE_*: knobs that change how the kernels run, never what they return."""
        _check(lib().rt_scene_set_tuning(self._h, C.c_int(key), C.c_int(value)))

    # ---- Sampler::sample_image: mean over opts.samples_per_pixel passes ----
    def render(self, camera, opts):
        n = C.c_uint64()
        _check(lib().rt_render_output_floats(C.byref(opts), C.byref(n)))
        out = np.zeros(n.value, dtype=np.float32)
        rays = C.c_uint64()
        _check(lib().rt_render(self._h, C.byref(camera), C.byref(opts), _p(out, C.c_float), C.byref(rays)))
        if opts.output_layout == abi.RT_LAYOUT_FRAME:
            out = out.reshape(opts.height, opts.width, 3)
        else:
            out = out.reshape(-1, 3)
        return out, rays.value

    def render_rgb8(self, camera, opts, gamma=2.2):
        """rt_render + the output stage on the device: the 8-bit image save_data_to_image would write (lib.rs:89-97)."""
        n = C.c_uint64()
        _check(lib().rt_render_output_floats(C.byref(opts), C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        rays = C.c_uint64()
        _check(lib().rt_render_rgb8(self._h, C.byref(camera), C.byref(opts), C.c_float(gamma), _p(out, C.c_uint8), C.byref(rays)))
        return (out.reshape(opts.height, opts.width, 3) if opts.output_layout == abi.RT_LAYOUT_FRAME else out.reshape(-1, 3)), rays.value

    def output_rgb8_device(self, d_rgb_ptr, n_values, d_out_ptr, gamma=2.2, stream=0):
        _check(lib().rt_output_rgb8_device(self._h, C.c_void_p(d_rgb_ptr), C.c_uint64(n_values), C.c_float(gamma), C.c_void_p(d_out_ptr),
                                           C.c_void_p(stream)))

    def render_device(self, camera, opts, d_out_ptr, d_rays_ptr=None, stream=0):
        """Asynchronous render into device memory (raw pointers, e.g. torch.Tensor.data_ptr())."""
        _check(lib().rt_render_device(self._h, C.byref(camera), C.byref(opts), C.c_void_p(d_out_ptr),
                                      C.c_void_p(d_rays_ptr) if d_rays_ptr else None, C.c_void_p(stream)))

    def gather_info(self):
        """(rt_gather_mode, why): how a multi-device scene moves its members' shards into devices[0] (rt_scene_gather_info)"""
        mode, note = C.c_int(), C.create_string_buffer(512)
        _check(lib().rt_scene_gather_info(self._h, C.byref(mode), note, C.c_uint64(512)))
        return mode.value, note.value.decode()

    def auto_sample_split(self, opts):
        """what opts.sample_split = 0 resolves to on this scene (rt_scene_auto_sample_split: the library's one rule)"""
        v = C.c_uint32()
        _check(lib().rt_scene_auto_sample_split(self._h, C.byref(opts), C.byref(v)))
        return v.value

    def last_kernel_ms(self):
        ms, n = C.c_float(), C.c_uint32()
        _check(lib().rt_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_launch_info(self):
        """What the most recent render launched (rt_launch_info) as a dict."""
        li = abi.LaunchInfo()
        _check(lib().rt_last_launch_info(self._h, C.byref(li)))
        d = {name: getattr(li, name) for name, _ in abi.LaunchInfo._fields_ if name not in ("kernel", "reserved")}
        d["kernel"] = li.kernel.decode()
        return d

    # ---- AccelerationStructure::check_hit / check_hit_index for batches ----
    def check_hit(self, origins, directions):
        rays = _pack_rays(origins, directions)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        _check(lib().rt_check_hit(self._h, rays.ctypes.data_as(C.POINTER(abi.RayDesc)), C.c_uint64(rays.shape[0]),
                                  out.ctypes.data_as(C.POINTER(abi.HitRecord))))
        return out

    def check_hit_index(self, origins, directions, indices):
        rays = _pack_rays(origins, directions)
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        _check(lib().rt_check_hit_index(self._h, rays.ctypes.data_as(C.POINTER(abi.RayDesc)), _p(idx, C.c_uint64),
                                        C.c_uint64(rays.shape[0]), out.ctypes.data_as(C.POINTER(abi.HitRecord))))
        return out


def output_floats(opts):
    n = C.c_uint64()
    _check(lib().rt_render_output_floats(C.byref(opts), C.byref(n)))
    return n.value


def shard_pixel_order(opts):
    n = output_floats(_with_layout(opts, abi.RT_LAYOUT_SHARD)) // 3
    out = np.zeros(max(1, n), dtype=np.uint64)
    _check(lib().rt_shard_pixel_order(C.byref(opts), _p(out, C.c_uint64), C.c_uint64(n)))
    return out[:n]


def _with_layout(opts, layout):
    o = abi.RenderOpts()
    C.memmove(C.byref(o), C.byref(opts), C.sizeof(o))
    o.output_layout = layout
    return o


class SamplerProgress:
    """samplers/mod.rs:49-63."""

    def __init__(self, pixel_num, channels=3):
        self.samples_completed = 0
        self.rays_shot = 0
        self.current_image = np.zeros(pixel_num * channels, dtype=np.float32)


class RandomSampler:
    """`impl Sampler` backed by the HIP kernels (samplers/random_sampler.rs:10-99).

    The reference hands its callback ONE image per pass; shipping 24.9 MB over PCIe per pass is
    what the batch ABI avoids, so passes are rendered `batch` at a time and the callback receives
    each batch's mean together with the number of passes it stands for.  With batch=1 the callback
    contract is exactly the reference's: f(data, progress, i) for i = 1..=spp, True cancels.
    """

    def __init__(self, batch=None):
        self.batch = batch

    def sample_image(self, render_options, camera, scene, presentation_update=None):
        """rt_sample_image: batch j+1 renders while batch j is copied out and handed to the callback."""
        opts = _with_layout(render_options, abi.RT_LAYOUT_FRAME)
        state = {"error": None}

        def trampoline(_data, p, done):
            try:
                progress = SamplerProgress(0)
                progress.samples_completed = int(p.contents.samples_completed)
                progress.rays_shot = int(p.contents.rays_shot)
                # a view of the sampler's pinned buffer: valid during the callback only
                progress.current_image = np.ctypeslib.as_array(p.contents.current_image, shape=(int(p.contents.n_floats),))
                if presentation_update is None:
                    return 0
                data, f = presentation_update
                return 1 if f(data, progress, int(done)) else 0
            except BaseException as e:  # an exception must not unwind through the C frames
                state["error"] = e
                return 1

        cb = abi.PresentationUpdate(trampoline)
        _check(lib().rt_sample_image(scene._h, C.byref(camera), C.byref(opts), C.c_uint64(self.batch or 0), cb, None))
        if state["error"] is not None:
            raise state["error"]


def running_mean_update(image, progress, i):
    """The TUI callback (src/main.rs:175-191) generalised to a batch of `progress.samples_completed`
    passes: image += (batch_mean - image) * n / i."""
    n = progress.samples_completed
    image += (progress.current_image - image) * (np.float32(n) / np.float32(i))
    return False


def output_rgb8(image, gamma=2.2):
    """save_data_to_image's pixel conversion (crates/output/src/lib.rs:92-95): (v.powf(1/gamma) * 255.999) as u8."""
    a = np.ascontiguousarray(image, dtype=np.float32)
    out = np.zeros(a.shape, dtype=np.uint8)
    _check(lib().rt_output_rgb8(_p(a, C.c_float), C.c_uint64(a.size), C.c_float(gamma), _p(out, C.c_uint8)))
    return out


def save_image(filename, image, gamma=2.2):
    """save_data_to_image (crates/output/src/lib.rs:74-113): .png .ppm .bmp .tiff (RGB8 after gamma) or .exr (floats) by extension."""
    a = np.ascontiguousarray(image, dtype=np.float32)
    h, w, _ = a.shape
    _check(lib().rt_output_save(filename.encode(), _p(a, C.c_float), C.c_uint32(w), C.c_uint32(h), C.c_float(gamma)))


def get_readable_duration(seconds):
    """output::get_readable_duration (crates/output/src/lib.rs:33-63), whole seconds as Duration::as_secs."""
    total = int(seconds)
    days, hours, minutes, secs = total // 86400, (total % 86400) // 3600, (total % 3600) // 60, total % 60

    def part(n, unit):
        return "" if n == 0 else (f"{n} {unit}, " if n == 1 else f"{n} {unit}s, ")

    tail = "~0 seconds" if secs == 0 else (f"{secs} second" if secs == 1 else f"{secs} seconds")
    return part(days, "day") + part(hours, "hour") + part(minutes, "minute") + tail


def final_statistics(seconds, ray_count, samples):
    """The message of output::print_final_statistics (crates/output/src/lib.rs:115-124)."""
    return (f"Finished rendering:\n\tSamples:\t{samples}\n\tTime taken:\t{get_readable_duration(seconds)}\n"
            f"\tRays shot:\t{ray_count} @ {ray_count / seconds / 1000000.0:.2f} Mray/s")
