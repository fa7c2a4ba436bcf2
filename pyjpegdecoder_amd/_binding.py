"""ctypes binding of libmijpeg.so (include/mijpeg.h).  Fails loudly when the HIP library is missing or
no GPU is present — there is no CPU fallback in the product path."""
from __future__ import annotations

import ctypes
from pathlib import Path
from typing import Optional

import numpy as np

from .errors import BackendError

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libmijpeg.so"

MJ_OK, MJ_ERR_INVALID, MJ_ERR_HIP, MJ_ERR_UNSUPPORTED = 0, -1, -2, -3
MJ_ST_OK, MJ_ST_BAD_CODE, MJ_ST_OVERRUN, MJ_ST_DESYNC, MJ_ST_TAIL, MJ_ST_UNCONVERGED, MJ_ST_INTERNAL = 0, 1, 2, 3, 4, 5, 6
MJ_MEM_NONE, MJ_MEM_HOST, MJ_MEM_DEVICE = 0, 1, 2
MJ_LAYOUT_XMAJOR, MJ_LAYOUT_ROWMAJOR, MJ_LAYOUT_PLANAR_XMAJOR, MJ_LAYOUT_PLANAR_ROWMAJOR = 0, 1, 2, 3
MJ_FLAG_KEEP_COEF, MJ_FLAG_KEEP_PLANES, MJ_FLAG_KEEP_IDCT, MJ_FLAG_EXACT_ONLY, MJ_FLAG_SPEC_REFINE = 1, 2, 4, 8, 16
MJ_FLAG_GPU_SEGMENT = 32
MJ_FLAG_NO_SYNC = 64

# every symbol include/mijpeg.h declares (tests check the library exports all of them)
EXPORTS = (
    "mj_create", "mj_destroy", "mj_last_error", "mj_version", "mj_context_wait_event",
    "mj_plan_create", "mj_plan_destroy", "mj_plan_get_info", "mj_plan_image_offsets",
    "mj_plan_execute", "mj_plan_execute_stage1", "mj_plan_execute_stage2", "mj_plan_sync",
    "mj_plan_device_buffers", "mj_plan_read", "mj_plan_write_coef", "mj_plan_fill_coef",
    "mj_decode_baseline_batch", "mj_idct_batch", "mj_plan_time_stages", "mj_plan_time_execute", "mj_plan_idct_levels", "mj_host_idct_table", "mj_host_assemble", "mj_plan_stage1_form", "mj_set_option", "mj_get_option", "mj_debug_stage1_form", "mj_debug_fused_shape", "mj_debug_count_tables",
    "mj_device_copy_rate", "mj_context_launch_clock", "mj_debug_prog_split", "mj_debug_fused_applies", "mj_plan_tune_placement",
)
MJ_FORM_WAVE, MJ_FORM_LANES, MJ_FORM_SYNC, MJ_FORM_SCANS, MJ_FORM_WG_TABLES, MJ_FORM_RESOLVED, MJ_FORM_FUSED, MJ_FORM_COUNT_RESOLVED = 0, 1, 2, 3, 16, 32, 64, 128
MJ_HOST_DECLINED = 1


class HuffSpecC(ctypes.Structure):
    _fields_ = [("bits", ctypes.c_uint8 * 16), ("vals", ctypes.c_uint8 * 256)]


class ImageDescC(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ncomp", ctypes.c_int32),
                ("hs", ctypes.c_int32 * 3), ("vs", ctypes.c_int32 * 3), ("qt_sel", ctypes.c_int32 * 3),
                ("dc_sel", ctypes.c_int32 * 3), ("ac_sel", ctypes.c_int32 * 3),
                ("restart_interval", ctypes.c_int32),
                ("mcu_count_h", ctypes.c_int32), ("mcu_count_v", ctypes.c_int32),
                ("n_segments", ctypes.c_int32), ("first_segment", ctypes.c_int64)]


class ScanDescC(ctypes.Structure):
    _fields_ = [("image", ctypes.c_int32), ("n_comp", ctypes.c_int32), ("comp", ctypes.c_int32 * 3),
                ("dc_sel", ctypes.c_int32 * 3), ("ac_sel", ctypes.c_int32 * 3),
                ("ss", ctypes.c_int32), ("se", ctypes.c_int32), ("ah", ctypes.c_int32), ("al", ctypes.c_int32),
                ("restart_interval", ctypes.c_int32), ("mcu_count_h", ctypes.c_int32), ("mcu_count_v", ctypes.c_int32),
                ("n_segments", ctypes.c_int32), ("first_segment", ctypes.c_int64)]


class BatchC(ctypes.Structure):
    _fields_ = [("n_images", ctypes.c_int32), ("images", ctypes.POINTER(ImageDescC)),
                ("blob", ctypes.c_void_p), ("blob_len", ctypes.c_int64), ("blob_mem", ctypes.c_int32),
                ("n_segments", ctypes.c_int64), ("seg_begin", ctypes.c_void_p), ("seg_end", ctypes.c_void_p),
                ("n_huff", ctypes.c_int32), ("huff", ctypes.POINTER(HuffSpecC)),
                ("n_qt", ctypes.c_int32), ("qt", ctypes.c_void_p),
                ("layout", ctypes.c_int32), ("flags", ctypes.c_uint32),
                ("n_scans", ctypes.c_int32), ("scans", ctypes.POINTER(ScanDescC))]


class HostJobC(ctypes.Structure):
    _fields_ = [("n_files", ctypes.c_int32), ("files", ctypes.POINTER(ctypes.c_char_p)), ("sizes", ctypes.c_void_p),
                ("file_off", ctypes.c_void_p), ("blob", ctypes.c_void_p), ("blob_len", ctypes.c_int64),
                ("images", ctypes.POINTER(ImageDescC)), ("seg_begin", ctypes.c_void_p), ("seg_end", ctypes.c_void_p),
                ("huff", ctypes.POINTER(HuffSpecC)), ("huff_cap", ctypes.c_int32),
                ("qt", ctypes.c_void_p), ("qt_cap", ctypes.c_int32), ("n_threads", ctypes.c_int32),
                ("n_huff", ctypes.c_int32), ("n_qt", ctypes.c_int32), ("declined_file", ctypes.c_int32),
                ("skip", ctypes.c_void_p), ("n_accepted", ctypes.c_int32)]


class PlanInfoC(ctypes.Structure):
    _fields_ = [("total_blocks", ctypes.c_int64), ("total_mcus", ctypes.c_int64), ("total_pixels", ctypes.c_int64),
                ("rgb_bytes", ctypes.c_int64), ("entropy_bytes", ctypes.c_int64)]


_lib = None


def load_library():
    """dlopen libmijpeg.so and declare the prototypes.  Raises BackendError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise BackendError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C pyjpegdecoder_amd/csrc`). There is no CPU fallback.")
    # PyTorch's ROCm wheels bundle their own HIP / HSA runtime under the same soname as /opt/rocm's.  Whichever is loaded first
    # serves the whole process: torch's first, and this library binds to it; /opt/rocm's first (through this library), and a
    # later `import torch` finds "No HIP GPUs are available".  So torch — which the device-tensor API needs anyway — goes first.
    try:
        import torch  # noqa: F401
    except Exception:      # no PyTorch: the host-array API works on /opt/rocm's runtime alone
        pass
    try:
        L = ctypes.CDLL(str(LIB_PATH))
    except OSError as exc:
        raise BackendError(f"cannot load {LIB_PATH}: {exc}") from exc
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    L.mj_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.mj_destroy.argtypes = [vp]
    L.mj_destroy.restype = None
    L.mj_last_error.argtypes = [vp]
    L.mj_last_error.restype = ctypes.c_char_p
    L.mj_context_wait_event.argtypes = [vp, vp]
    L.mj_plan_create.argtypes = [vp, ctypes.POINTER(BatchC), ctypes.POINTER(vp)]
    L.mj_plan_destroy.argtypes = [vp]
    L.mj_plan_destroy.restype = None
    L.mj_plan_get_info.argtypes = [vp, ctypes.POINTER(PlanInfoC)]
    L.mj_plan_image_offsets.argtypes = [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(i64)]
    L.mj_plan_execute.argtypes = [vp, vp, vp]
    L.mj_plan_execute_stage1.argtypes = [vp, vp]
    L.mj_plan_execute_stage2.argtypes = [vp, vp, vp]
    L.mj_plan_sync.argtypes = [vp]
    L.mj_plan_device_buffers.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.mj_plan_read.argtypes = [vp, vp, vp, vp, vp, vp]
    L.mj_plan_write_coef.argtypes = [vp, vp, i32]
    L.mj_decode_baseline_batch.argtypes = [vp, ctypes.POINTER(BatchC), vp, vp, vp]
    L.mj_idct_batch.argtypes = [vp, ctypes.POINTER(BatchC), vp, vp]
    L.mj_plan_time_stages.argtypes = [vp, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.mj_plan_fill_coef.argtypes = [vp, ctypes.c_int]
    L.mj_plan_time_execute.argtypes = [vp, ctypes.c_int, vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.mj_host_idct_table.argtypes = [vp]
    L.mj_host_idct_table.restype = None
    L.mj_host_assemble.argtypes = [ctypes.POINTER(HostJobC)]
    L.mj_plan_stage1_form.argtypes = [vp]
    L.mj_plan_idct_levels.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.mj_set_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.mj_get_option.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32]
    L.mj_debug_stage1_form.argtypes = [vp, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_int32,
                                       ctypes.POINTER(ctypes.c_int32)]
    _lib = L
    return L


def stage1_form_rule(seg_len, blob_len=None, n_huff=4, traits=0, force=None, forced_chunk=0):
    """mj_debug_stage1_form (host only): (MJ_FORM_*, chunk bytes, chunks, dealt out by length) for a batch whose restart
    segments have these byte lengths — the rule mj_plan_create applies (csrc/form_select.h)."""
    a = np.ascontiguousarray(seg_len, dtype=np.int32)
    out = (ctypes.c_int32 * 4)()
    rc = load_library().mj_debug_stage1_form(_ptr(a), a.size, int(a.sum()) + 4096 if blob_len is None else blob_len, n_huff, traits,
                                             force.encode() if force else None, forced_chunk, out)
    if rc != MJ_OK:
        raise ValueError("mj_debug_stage1_form: bad arguments")
    return int(out[0]), int(out[1]), int(out[2]), bool(out[3])


def count_tables(huff_specs, roles, wbits=12):
    """mj_debug_count_tables (host only): (words, tab_bytes) — the synchronisation form's counting tables for these tables
    (list of (bits[16], vals) pairs; roles: 1 = DC, 2 = AC) — or None where such a batch takes the classic rounds."""
    L = load_library()
    n = len(huff_specs)
    arr = (HuffSpecC * n)()
    for i, (bits, vals) in enumerate(huff_specs):
        for j in range(16):
            arr[i].bits[j] = int(bits[j])
        for j, v in enumerate(vals):
            arr[i].vals[j] = int(v)
    r = (ctypes.c_int32 * n)(*[int(x) for x in roles])
    tb = ctypes.c_int32(0)
    L.mj_debug_count_tables.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64,
                                        ctypes.POINTER(ctypes.c_int32)]
    rc = L.mj_debug_count_tables(ctypes.byref(arr), n, ctypes.byref(r), wbits, None, 0, ctypes.byref(tb))
    if rc == MJ_ERR_UNSUPPORTED:
        return None
    if rc != MJ_OK:
        raise ValueError("mj_debug_count_tables: bad arguments")
    out = np.zeros(n * tb.value // 4, dtype=np.uint32)
    rc = L.mj_debug_count_tables(ctypes.byref(arr), n, ctypes.byref(r), wbits, _ptr(out), out.size, ctypes.byref(tb))
    if rc != MJ_OK:
        raise ValueError("mj_debug_count_tables: bad arguments")
    return out, tb.value


def fused_shape_rule(n_images, segments_per_image, hmax=2, vmax=2, transposed=False, cus=256, n_ac=2, n_dc=2, ac_slot_bytes=17024,
                     want_consumers=8):
    """mj_debug_fused_shape (host only): dict(ok, images_per_wg [and pass], producers, lanes, consumers, producer_lds, passes,
    workgroups) of a fused launch."""
    out = (ctypes.c_int32 * 8)()
    L = load_library()
    L.mj_debug_fused_shape.argtypes = [ctypes.c_int32] * 10 + [ctypes.POINTER(ctypes.c_int32)]
    if L.mj_debug_fused_shape(cus, n_ac, n_dc, ac_slot_bytes, hmax, vmax, int(transposed), n_images, segments_per_image, want_consumers, out) != MJ_OK:
        raise ValueError("mj_debug_fused_shape: bad arguments")
    return dict(zip(("ok", "images_per_wg", "producers", "lanes", "consumers", "producer_lds", "passes", "workgroups"),
                    (bool(out[0]),) + tuple(int(x) for x in out[1:])))


def fused_applies_rule(layout, hmax, vmax, mcus_per_row, mcu_rows, restart_interval, n_images=1024, ncomp=3, traits=0, flags=0):
    """mj_debug_fused_applies (host only): 0 the two launches, 1 one fused launch, 2 fused with the hand-off across workgroups."""
    L = load_library()
    out = ctypes.c_int32(-1)
    L.mj_debug_fused_applies.argtypes = [ctypes.c_int32] * 8 + [ctypes.c_uint32] * 2 + [ctypes.POINTER(ctypes.c_int32)]
    if L.mj_debug_fused_applies(layout, ncomp, hmax, vmax, mcus_per_row, mcu_rows, restart_interval, n_images, traits, flags, ctypes.byref(out)) != MJ_OK:
        raise ValueError("mj_debug_fused_applies: bad arguments")
    return out.value


def prog_split_rule(n_images, scans, mode=1, n_bands=68, wave_slots=0, parts=0):
    """mj_debug_prog_split (host only): (split flags per scan, parts per band) for a progressive batch whose scans are
    (image, restart segments, entropy-coded bytes or -1 where the scan is no refining AC scan of one component)."""
    n = len(scans)
    img = np.asarray([s[0] for s in scans], dtype=np.int32)
    seg = np.asarray([s[1] for s in scans], dtype=np.int32)
    byt = np.asarray([s[2] for s in scans], dtype=np.int64)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    po = ctypes.c_int32()
    L = load_library()
    L.mj_debug_prog_split.argtypes = [ctypes.c_int32] * 6 + [ctypes.c_void_p] * 4 + [ctypes.POINTER(ctypes.c_int32)]
    if L.mj_debug_prog_split(mode, n_images, n_bands, wave_slots, parts, n, _ptr(img), _ptr(seg), _ptr(byt), _ptr(out), ctypes.byref(po)) != MJ_OK:
        raise ValueError("mj_debug_prog_split: bad arguments")
    return out[:n].astype(bool).tolist(), int(po.value)


class UnknownOption(ValueError):
    """mj_set_option / mj_get_option: no such option in this library."""


def get_option(name: str) -> str:
    """mj_get_option: the value an option holds, "" = the default."""
    buf = ctypes.create_string_buffer(64)
    if load_library().mj_get_option(name.encode(), buf, 64) != MJ_OK:
        raise UnknownOption(f"libmijpeg has no option {name!r}")
    return buf.value.decode()


def set_option(name: str, value=None):
    """mj_set_option: a test / tuning switch of the library, process-wide (the library does not read the environment for
    these).  value None = back to the default.  UnknownOption for a name the library does not have, ValueError for a value
    outside the option's range (the option keeps what it had)."""
    get_option(name)
    rc = load_library().mj_set_option(name.encode(), None if value is None else str(value).encode())
    if rc != MJ_OK:
        raise ValueError(f"libmijpeg: {value!r} is outside what option {name} takes (include/mijpeg.h)")


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Context:
    """mj_context: one per GPU per thread."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = ctypes.c_void_p()
        rc = self.lib.mj_create(device, ctypes.byref(h))
        if rc != MJ_OK:
            raise BackendError(self.lib.mj_last_error(None).decode() or f"mj_create failed ({rc})")
        self.handle = h
        self.device = device

    def check(self, rc: int):
        if rc != MJ_OK:
            msg = self.lib.mj_last_error(self.handle).decode()
            if rc == MJ_ERR_UNSUPPORTED:
                from .errors import UnsupportedJpeg
                raise UnsupportedJpeg(msg)
            raise BackendError(f"libmijpeg error {rc}: {msg}")

    def wait_event(self, hip_event: int):
        """Everything queued on the context's stream from now on runs after `hip_event` (a hipEvent_t handle, e.g.
        ``torch.cuda.Event.cuda_event``) has happened."""
        self.check(self.lib.mj_context_wait_event(self.handle, hip_event))

    def copy_rate_gbs(self, nbytes: int = 1 << 31, iters: int = 5) -> float:
        """mj_device_copy_rate: GB/s (read + written) of a plain 16-bytes-per-lane device copy of `nbytes`."""
        ms = ctypes.c_float()
        self.lib.mj_device_copy_rate.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        self.check(self.lib.mj_device_copy_rate(self.handle, int(nbytes), int(iters), ctypes.byref(ms)))
        return 2.0 * (int(nbytes) & ~15) / (ms.value * 1e-3) / 1e9

    def launch_clock(self):
        """mj_context_launch_clock: (shader MHz held during the latest fused launch, that launch's ms as workgroup 0 saw it)."""
        mhz, ms = ctypes.c_float(), ctypes.c_float()
        self.lib.mj_context_launch_clock.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
        self.check(self.lib.mj_context_launch_clock(self.handle, ctypes.byref(mhz), ctypes.byref(ms)))
        return mhz.value, ms.value

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mj_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """mj_plan over a prepared batch (see batch.PreparedBatch)."""

    def __init__(self, ctx: Context, batch_c: BatchC, keepalive):
        self.ctx = ctx
        self._keep = keepalive
        h = ctypes.c_void_p()
        ctx.check(ctx.lib.mj_plan_create(ctx.handle, ctypes.byref(batch_c), ctypes.byref(h)))
        self.handle = h
        info = PlanInfoC()
        ctx.check(ctx.lib.mj_plan_get_info(h, ctypes.byref(info)))
        self.info = info

    def execute(self, stream: int = 0, rgb_device: int = 0):
        self.ctx.check(self.ctx.lib.mj_plan_execute(self.handle, stream or None, rgb_device or None))

    def execute_stage1(self, stream: int = 0):
        self.ctx.check(self.ctx.lib.mj_plan_execute_stage1(self.handle, stream or None))

    def execute_stage2(self, stream: int = 0, rgb_device: int = 0):
        self.ctx.check(self.ctx.lib.mj_plan_execute_stage2(self.handle, stream or None, rgb_device or None))

    def sync(self):
        self.ctx.check(self.ctx.lib.mj_plan_sync(self.handle))

    def stage1_form(self) -> int:
        """MJ_FORM_* (| MJ_FORM_WG_TABLES): which stage-1 form the library chose for this batch."""
        return int(self.ctx.lib.mj_plan_stage1_form(self.handle))

    def write_coef(self, coef: np.ndarray):
        coef = np.ascontiguousarray(coef, dtype=np.int16)
        assert coef.size == self.info.total_blocks * 64
        self.ctx.check(self.ctx.lib.mj_plan_write_coef(self.handle, _ptr(coef), MJ_MEM_HOST))

    def fill_coef(self, byte_value: int):
        """Test hook (mj_plan_fill_coef): poison the coefficient store."""
        self.ctx.check(self.ctx.lib.mj_plan_fill_coef(self.handle, int(byte_value)))

    def read(self, rgb=True, coef=False, planes=False, idct=False):
        out = {}
        a_rgb = np.empty(self.info.rgb_bytes, dtype=np.uint8) if rgb else None
        a_coef = np.empty((self.info.total_blocks, 64), dtype=np.int16) if coef else None
        a_pl = np.empty(self.info.rgb_bytes, dtype=np.int16) if planes else None
        a_id = np.empty((self.info.total_blocks, 64), dtype=np.int16) if idct else None
        n_images = self._keep["n_images"]
        st = np.zeros(n_images, dtype=np.int32)
        self.ctx.check(self.ctx.lib.mj_plan_read(self.handle, _ptr(a_rgb), _ptr(a_coef), _ptr(a_pl), _ptr(a_id), _ptr(st)))
        out.update(rgb=a_rgb, coef=a_coef, planes=a_pl, idct=a_id, status=st)
        return out

    def image_offsets(self, i: int):
        b, r = ctypes.c_int64(), ctypes.c_int64()
        self.ctx.check(self.ctx.lib.mj_plan_image_offsets(self.handle, i, ctypes.byref(b), ctypes.byref(r)))
        return b.value, r.value

    def device_buffers(self):
        c, r, p, d = (ctypes.c_void_p() for _ in range(4))
        self.ctx.check(self.ctx.lib.mj_plan_device_buffers(self.handle, ctypes.byref(c), ctypes.byref(r), ctypes.byref(p), ctypes.byref(d)))
        return {"coef": c.value, "rgb": r.value, "planes": p.value, "idct": d.value}

    def idct_levels(self):
        """(blocks, sent on by the fp32 level, sent on by the fp64 level) of the latest stage-2 execute with seam outputs."""
        c = (ctypes.c_uint64 * 3)()
        self.ctx.check(self.ctx.lib.mj_plan_idct_levels(self.handle, c))
        return int(c[0]), int(c[1]), int(c[2])

    def time_stages(self, iters: int = 10, rgb_device: int = 0):
        s1, s2 = ctypes.c_float(), ctypes.c_float()
        self.ctx.check(self.ctx.lib.mj_plan_time_stages(self.handle, iters, rgb_device or None, ctypes.byref(s1), ctypes.byref(s2)))
        return s1.value, s2.value

    def time_execute(self, iters: int = 10, rgb_device: int = 0):
        """(front_ms, main_ms) of the launches execute() makes: a fused plan's stage 0 and fused launch, else the two stages."""
        f, m = ctypes.c_float(), ctypes.c_float()
        self.ctx.check(self.ctx.lib.mj_plan_time_execute(self.handle, iters, rgb_device or None, ctypes.byref(f), ctypes.byref(m)))
        return f.value, m.value

    def tune_placement(self, stream: int = 0, rgb_device: int = 0, candidates: int = 4):
        """mj_plan_tune_placement: tries `candidates` coefficient stores for a fused plan that will be executed many times into
        `rgb_device` (then as many stage-0 stream buffers), keeps the fastest.  Returns (ms per execute of every store tried, index
        of the one that stayed); `best_ms` afterwards = ms per execute with what the plan ended up with."""
        ms = (ctypes.c_float * candidates)()
        chosen, best = ctypes.c_int32(), ctypes.c_float()
        self.ctx.lib.mj_plan_tune_placement.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                        ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)]
        self.ctx.check(self.ctx.lib.mj_plan_tune_placement(self.handle, stream or None, rgb_device or None, candidates, ms, ctypes.byref(chosen),
                                                           ctypes.byref(best)))
        self.best_ms = float(best.value)
        return [float(x) for x in ms], int(chosen.value)

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.mj_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
