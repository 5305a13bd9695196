"""sdf_playground_amd -- MI355X-native SDF raymarch renderer (hot path of Gotbread/sdf-playground).

Python host side above the C ABI (include/sdfr.h, libsdfr.so).  The classes mirror the
reference's host interface for this path so that code written against the reference reads
the same here:

    SDFRenderer   Engine/SDFRenderer.h:17-44   init / initShader / setParameters /
                                               getVariableMap / render
    Camera        Engine/Camera.h:5-64         SetEye / SetLookat / SetDirection / SetAspect /
                                               SetFOVY / SetRoll  (FPS mode)
    Variable      Engine/ShaderVariable.h:6-12 minval / maxval / start / step / value

All pixels come from the HIP kernels; if libsdfr.so cannot be loaded or no GPU is present
the calls raise -- there is no CPU fallback.
"""
import ctypes
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDFR_LIBRARY") or os.path.join(_HERE, "libsdfr.so")  # SDFR_LIBRARY: developer builds (tools/phase_clocks.py)

SDFR_OK = 0
SCHEDULE_WAVEFRONT = 0
SCHEDULE_PIXEL = 1
LAUNCH_AUTO, LAUNCH_PER_TILE, LAUNCH_PERSISTENT = 0, 1, 2
RGBA32F = 0
RGBA16F = 1
STRIP_RGB32F_A8 = 2  # strips only: rgb float triples + one flag byte per pixel (lossless, 13 B/pixel)
STRIP_RGB16F_A8 = 3  # strips only: rgb half triples + one flag byte per pixel (the reference's RGBA16F target in 7 B/pixel)
COMM_ID_BYTES = 128
STRIP_ROWS = 8

_STATUS = {
    0: "SDFR_OK", -1: "SDFR_ERR_INVALID_ARGUMENT", -2: "SDFR_ERR_UNKNOWN_SCENE", -3: "SDFR_ERR_UNKNOWN_VARIABLE",
    -4: "SDFR_ERR_NO_SCENE", -5: "SDFR_ERR_HIP", -6: "SDFR_ERR_NO_DEVICE", -7: "SDFR_ERR_COMPILE", -8: "SDFR_ERR_COMM", -9: "SDFR_ERR_INTERNAL",
}


class SdfrError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("%s: %s" % (_STATUS.get(code, str(code)), message))
        self.code = code


class _CVariable(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 48), ("minval", ctypes.c_float), ("maxval", ctypes.c_float), ("start", ctypes.c_float),
                ("step", ctypes.c_float), ("value", ctypes.c_float)]


class _CTiming(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("ms", ctypes.c_double)]


class Limits(ctypes.Structure):
    """sdfr_limits: the driver's compile-time limits (pshader_sdf.hlsl:60-64,350) at run time."""

    _fields_ = [("iter_count", ctypes.c_int), ("bounce_count", ctypes.c_int), ("ray_count", ctypes.c_int), ("light_count", ctypes.c_int),
                ("range", ctypes.c_float), ("max_cost_default", ctypes.c_int), ("extension_lights", ctypes.c_int),
                ("extension_marble_reflection", ctypes.c_float),
                # pshader_sdf.hlsl:31-35 at run time; defaults 1e-4, 1e-4, 1e-3, 1e-3, 3e-4 (anything else: extension)
                ("dist_eps", ctypes.c_float), ("grad_eps", ctypes.c_float), ("reflect_eps", ctypes.c_float), ("refract_eps", ctypes.c_float),
                ("shadow_eps", ctypes.c_float)]


class Stats(ctypes.Structure):
    _fields_ = [("ms_gpu", ctypes.c_double), ("ms_march", ctypes.c_double), ("ms_shade", ctypes.c_double), ("pixels", ctypes.c_uint64),
                ("rays", ctypes.c_uint64), ("march_evals", ctypes.c_uint64), ("hits", ctypes.c_uint64), ("march_launches", ctypes.c_uint32),
                ("shade_launches", ctypes.c_uint32)]


# every symbol include/sdfr.h declares (tests check that the library exports all of them)
EXPORTED_SYMBOLS = [
    "sdfr_create", "sdfr_destroy", "sdfr_last_error", "sdfr_set_stream", "sdfr_scene_count", "sdfr_scene_name", "sdfr_load_scene",
    "sdfr_current_scene", "sdfr_var_count", "sdfr_var_info", "sdfr_var_set", "sdfr_var_get", "sdfr_vars_reset", "sdfr_set_camera",
    "sdfr_set_camera_lookat", "sdfr_set_camera_direction", "sdfr_get_camera", "sdfr_set_time", "sdfr_get_limits", "sdfr_set_limits",
    "sdfr_set_schedule", "sdfr_set_profiling", "sdfr_strip_buffer_pixels", "sdfr_render", "sdfr_render_strips", "sdfr_assemble_strips",
    "sdfr_sync", "sdfr_set_frames_in_flight", "sdfr_wait_frame", "sdfr_get_stats", "sdfr_selftest_math", "sdfr_selftest_exception", "sdfr_postprocess", "sdfr_load_scene_source", "sdfr_check_scene_source", "sdfr_load_scene_hlsl", "sdfr_check_scene_hlsl", "sdfr_translate_scene_hlsl", "sdfr_get_timings", "sdfr_strip_buffer_bytes",
    "sdfr_set_strip_split", "sdfr_strip_buffer_pixels_split", "sdfr_strip_buffer_bytes_split", "sdfr_render_private_strips",
    "sdfr_comm_unique_id", "sdfr_comm_create", "sdfr_comm_create_all", "sdfr_comm_destroy", "sdfr_comm_close", "sdfr_comm_library_info", "sdfr_comm_rank", "sdfr_comm_world",
    "sdfr_comm_last_error", "sdfr_comm_selftest", "sdfr_render_gather", "sdfr_render_gather_all", "sdfr_set_launch_mode", "sdfr_set_step_shortcuts",
    "sdfr_register_host_target",
]

_lib = None


def build(force=False):
    """Compile the HIP sources for gfx950 into libsdfr.so (in-tree)."""
    from . import buildlib as _build

    return _build.build(force=force)


def load_library():
    """dlopen libsdfr.so.  torch (when installed) is imported first so that both share one
    HIP runtime (same SONAME, libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SdfrError(-5, "libsdfr.so is not built: run `python -m sdf_playground_amd.buildlib` (needs hipcc)")
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    L.sdfr_create.argtypes = [ci, ctypes.POINTER(vp)]
    L.sdfr_destroy.argtypes = [vp]
    L.sdfr_destroy.restype = None
    L.sdfr_last_error.argtypes = [vp]
    L.sdfr_last_error.restype = ctypes.c_char_p
    L.sdfr_set_stream.argtypes = [vp, vp]
    L.sdfr_scene_name.argtypes = [ci]
    L.sdfr_scene_name.restype = ctypes.c_char_p
    L.sdfr_load_scene.argtypes = [vp, ctypes.c_char_p]
    L.sdfr_load_scene_source.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    L.sdfr_check_scene_source.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.sdfr_load_scene_hlsl.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    L.sdfr_check_scene_hlsl.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.sdfr_translate_scene_hlsl.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t]
    L.sdfr_current_scene.argtypes = [vp]
    L.sdfr_current_scene.restype = ctypes.c_char_p
    L.sdfr_var_count.argtypes = [vp]
    L.sdfr_var_info.argtypes = [vp, ci, ctypes.POINTER(_CVariable)]
    L.sdfr_var_set.argtypes = [vp, ctypes.c_char_p, cf]
    L.sdfr_var_get.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(cf)]
    L.sdfr_vars_reset.argtypes = [vp]
    f3 = ctypes.POINTER(cf)
    L.sdfr_set_camera.argtypes = [vp, f3, f3, f3, f3]
    L.sdfr_set_camera_lookat.argtypes = [vp, f3, f3, cf, cf, cf]
    L.sdfr_set_camera_direction.argtypes = [vp, f3, f3, cf, cf, cf]
    L.sdfr_get_camera.argtypes = [vp, f3]
    L.sdfr_set_time.argtypes = [vp, cf]
    L.sdfr_get_limits.argtypes = [vp, ctypes.POINTER(Limits)]
    L.sdfr_set_limits.argtypes = [vp, ctypes.POINTER(Limits)]
    L.sdfr_set_schedule.argtypes = [vp, ci]
    L.sdfr_set_profiling.argtypes = [vp, ci]
    L.sdfr_set_launch_mode.argtypes = [vp, ci]
    L.sdfr_set_step_shortcuts.argtypes = [vp, ci]
    L.sdfr_register_host_target.argtypes = [vp, vp, ctypes.c_size_t]
    L.sdfr_strip_buffer_pixels.argtypes = [ci, ci, ci]
    L.sdfr_strip_buffer_pixels.restype = ctypes.c_int64
    L.sdfr_strip_buffer_bytes.argtypes = [ci, ci, ci, ci]
    L.sdfr_strip_buffer_bytes.restype = ctypes.c_int64
    L.sdfr_set_strip_split.argtypes = [vp, ci, ci]
    L.sdfr_strip_buffer_pixels_split.argtypes = [ci, ci, ci, ci, ci]
    L.sdfr_strip_buffer_pixels_split.restype = ctypes.c_int64
    L.sdfr_strip_buffer_bytes_split.argtypes = [ci, ci, ci, ci, ci, ci]
    L.sdfr_strip_buffer_bytes_split.restype = ctypes.c_int64
    L.sdfr_render_private_strips.argtypes = [vp, ci, ci, vp, ci]
    L.sdfr_render.argtypes = [vp, ci, ci, vp, ci, ci, vp]
    L.sdfr_render_strips.argtypes = [vp, ci, ci, ci, ci, vp, ci]
    L.sdfr_assemble_strips.argtypes = [vp, ci, ci, ci, vp, vp, ci]
    L.sdfr_sync.argtypes = [vp]
    L.sdfr_set_frames_in_flight.argtypes = [vp, ci]
    L.sdfr_wait_frame.argtypes = [vp, vp]
    L.sdfr_get_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    L.sdfr_get_timings.argtypes = [vp, ctypes.POINTER(_CTiming), ci]
    L.sdfr_postprocess.argtypes = [vp, ci, ci, vp, vp, vp]
    L.sdfr_selftest_math.argtypes = [vp, ci, cf, ctypes.POINTER(ctypes.c_uint64)]
    L.sdfr_selftest_exception.argtypes = [vp, ci]
    L.sdfr_comm_unique_id.argtypes = [vp]
    L.sdfr_comm_create.argtypes = [vp, ci, ci, ci, ctypes.POINTER(vp)]
    L.sdfr_comm_create_all.argtypes = [ctypes.POINTER(ci), ci, ctypes.POINTER(vp)]
    L.sdfr_comm_destroy.argtypes = [vp]
    L.sdfr_comm_destroy.restype = None
    L.sdfr_comm_close.argtypes = [vp]
    L.sdfr_comm_library_info.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.sdfr_comm_rank.argtypes = [vp]
    L.sdfr_comm_world.argtypes = [vp]
    L.sdfr_comm_last_error.argtypes = [vp]
    L.sdfr_comm_last_error.restype = ctypes.c_char_p
    L.sdfr_comm_selftest.argtypes = [vp, ctypes.c_size_t, vp]
    L.sdfr_render_gather.argtypes = [vp, vp, ci, ci, vp, ci, ci]
    L.sdfr_render_gather_all.argtypes = [ctypes.POINTER(vp), ctypes.POINTER(vp), ci, ci, ci, vp, ci, ci]
    _lib = L
    return L


def scene_names():
    L = load_library()
    return [L.sdfr_scene_name(i).decode() for i in range(L.sdfr_scene_count())]


def check_scene_source(source, arch="gfx950"):
    """Compile a run-time scene without a device; returns (ok, compiler messages)."""
    if os.path.exists(source):
        with open(source) as f:
            source = f.read()
    log = ctypes.create_string_buffer(1 << 16)
    rc = load_library().sdfr_check_scene_source(source.encode(), arch.encode(), log, len(log))
    return rc == SDFR_OK, log.value.decode(errors="replace")


def check_scene_hlsl(source, arch="gfx950"):
    """Compile a scene written in the reference's dialect (an .hlsl scene file's text, or its path) without a device;
    returns (ok, compiler messages).  sdfr_check_scene_hlsl."""
    if os.path.exists(source):
        with open(source) as f:
            source = f.read()
    log = ctypes.create_string_buffer(1 << 16)
    rc = load_library().sdfr_check_scene_hlsl(source.encode(), arch.encode(), log, len(log))
    return rc == SDFR_OK, log.value.decode(errors="replace")


def translate_scene_hlsl(source):
    """The C++ the library generates from a scene in the reference's dialect (sdfr_translate_scene_hlsl): `struct UserScene`
    + the `Scene` typedef, to stand inside namespace sdfr after #include "sdfr_hlsl.h"."""
    if os.path.exists(source):
        with open(source) as f:
            source = f.read()
    L = load_library()
    n = L.sdfr_translate_scene_hlsl(source.encode(), None, 0)
    buf = ctypes.create_string_buffer(n)
    L.sdfr_translate_scene_hlsl(source.encode(), buf, n)
    return buf.value.decode()


def strip_buffer_pixels(width, height, world, split=(0, 1)):
    return int(load_library().sdfr_strip_buffer_pixels_split(width, height, world, split[0], split[1]))


def strip_buffer_bytes(width, height, world, fmt, split=(0, 1)):
    """Bytes of one rank's compact strip buffer in format `fmt` (RGBA32F, RGBA16F or STRIP_RGB32F_A8);
    split = (priv_count, priv_period) of setStripSplit."""
    return int(load_library().sdfr_strip_buffer_bytes_split(width, height, world, fmt, split[0], split[1]))


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


def to_radian(deg):
    """Math3D::ToRadian (Math3D.h:299-303) in fp32."""
    return float(np.float32(deg) * np.float32(3.14159265358979) / np.float32(180.0))


class Camera:
    """First-person camera of the reference (Engine/Camera.h), parameters only: the basis
    arithmetic runs in the C++ host library (sdfr_set_camera_lookat / _direction)."""

    def __init__(self):
        # Application.cpp:214-224
        self.eye = (0.0, 2.0, -3.0)
        self.target = (0.0, 1.0, 0.0)
        self.target_is_direction = False
        self.fovy = to_radian(60.0)
        self.aspect = float(np.float32(1200.0) / np.float32(800.0))
        self.roll = 0.0

    def SetEye(self, eye):
        self.eye = tuple(float(x) for x in eye)

    def SetLookat(self, lookat):
        self.target = tuple(float(x) for x in lookat)
        self.target_is_direction = False

    def SetDirection(self, direction):
        self.target = tuple(float(x) for x in direction)
        self.target_is_direction = True

    def SetAspect(self, aspect):
        self.aspect = float(aspect)

    def SetFOVY(self, fovy):
        self.fovy = float(fovy)

    def SetRoll(self, roll):
        self.roll = float(roll)


class Variable:
    """One shader variable; assigning .value updates the renderer (the reference's UI
    writes through a raw pointer into the map, VariableManager.cpp:105,157)."""

    def __init__(self, owner, name, c):
        self._owner, self.name = owner, name
        self.minval, self.maxval, self.start, self.step = c.minval, c.maxval, c.start, c.step

    @property
    def value(self):
        out = ctypes.c_float()
        self._owner._check(self._owner._L.sdfr_var_get(self._owner._h, self.name.encode(), ctypes.byref(out)))
        return out.value

    @value.setter
    def value(self, v):
        self._owner._check(self._owner._L.sdfr_var_set(self._owner._h, self.name.encode(), float(v)))

    def __repr__(self):
        return "Variable(%s: min=%g max=%g start=%g step=%g value=%g)" % (self.name, self.minval, self.maxval, self.start, self.step, self.value)


class SDFRenderer:
    """The SDF render stage.  Mirrors Engine/SDFRenderer.h:17-44."""

    def __init__(self, device=0):
        self._L = load_library()
        self._h = ctypes.c_void_p()
        self._stime = 0.0
        self.init(device)

    # bool init(Graphics&) -- here: bind to a GPU
    def init(self, device=0):
        if self._h:
            self._L.sdfr_destroy(self._h)
            self._h = ctypes.c_void_p()
        rc = self._L.sdfr_create(int(device), ctypes.byref(self._h))
        if rc != SDFR_OK:
            raise SdfrError(rc, "sdfr_create(device=%d) failed" % device)
        self.device = int(device)
        return True

    def close(self):
        if self._h:
            self._L.sdfr_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != SDFR_OK:
            raise SdfrError(rc, self._L.sdfr_last_error(self._h).decode())

    # bool initShader(ShaderIncluder&) with the scene substitution of Application::loadScene
    def initShader(self, scene):
        self._check(self._L.sdfr_load_scene(self._h, scene.encode()))
        return True

    loadScene = initShader

    # the reference's edit-and-reload: compile a scene from its source text (hiprtc).  `source`
    # is the text or a path to it; on a compile error the previous scene stays active and
    # SdfrError carries the compiler's messages (SceneManager.cpp:118-127)
    def initShaderSource(self, name, source):
        if os.path.exists(source):
            with open(source) as f:
                source = f.read()
        self._check(self._L.sdfr_load_scene_source(self._h, name.encode(), source.encode()))
        return True

    def initShaderHlsl(self, name, source):
        """A scene in the reference's own dialect: the text (or path) of an .hlsl scene file with map / map_normal / map_light /
        map_background, as Application::loadScene would substitute it into the shader (sdfr_load_scene_hlsl)."""
        if os.path.exists(source):
            with open(source) as f:
                source = f.read()
        self._check(self._L.sdfr_load_scene_hlsl(self._h, name.encode(), source.encode()))
        return True

    def currentScene(self):
        s = self._L.sdfr_current_scene(self._h)
        return s.decode() if s else None

    # void setParameters(float stime)
    def setParameters(self, stime):
        self._stime = float(stime)
        self._check(self._L.sdfr_set_time(self._h, self._stime))

    # VariableMap &getVariableMap(): ordered like std::map
    def getVariableMap(self):
        out = {}
        for i in range(self._L.sdfr_var_count(self._h)):
            c = _CVariable()
            self._check(self._L.sdfr_var_info(self._h, i, ctypes.byref(c)))
            out[c.name.decode()] = Variable(self, c.name.decode(), c)
        return out

    def setValue(self, name, value):
        """ShaderVariableManager::setValue: unknown names are ignored (returns False)."""
        rc = self._L.sdfr_var_set(self._h, name.encode(), float(value))
        if rc == -3:
            return False
        self._check(rc)
        return True

    def resetVariables(self):
        self._check(self._L.sdfr_vars_reset(self._h))

    def setCamera(self, camera):
        fn = self._L.sdfr_set_camera_direction if camera.target_is_direction else self._L.sdfr_set_camera_lookat
        self._check(fn(self._h, _f3(camera.eye), _f3(camera.target), camera.fovy, camera.aspect, camera.roll))

    def setCameraBasis(self, eye, front, right, top):
        """The raw constant-buffer form (SDFRenderer.h:29-34)."""
        self._check(self._L.sdfr_set_camera(self._h, _f3(eye), _f3(front), _f3(right), _f3(top)))

    def getCameraBasis(self):
        out = (ctypes.c_float * 12)()
        self._check(self._L.sdfr_get_camera(self._h, out))
        return np.array(out, np.float32).reshape(4, 3)

    def getLimits(self):
        l = Limits()
        self._check(self._L.sdfr_get_limits(self._h, ctypes.byref(l)))
        return l

    def setLimits(self, **kw):
        l = self.getLimits()
        for k, v in kw.items():
            if not hasattr(l, k):
                raise AttributeError(k)
            setattr(l, k, v)
        self._check(self._L.sdfr_set_limits(self._h, ctypes.byref(l)))

    def setSchedule(self, schedule):
        self._check(self._L.sdfr_set_schedule(self._h, int(schedule)))

    def setLaunchMode(self, mode):
        """LAUNCH_AUTO (the scene's own choice), LAUNCH_PER_TILE or LAUNCH_PERSISTENT (sdfr_set_launch_mode)."""
        self._check(self._L.sdfr_set_launch_mode(self._h, int(mode)))

    def setStepShortcuts(self, enabled):
        """sdfr_set_step_shortcuts: rays their scene knows to be misses already stop marching (same pixels, fewer steps
        counted); off = every step marched, step counters equal the reference's."""
        self._check(self._L.sdfr_set_step_shortcuts(self._h, 1 if enabled else 0))

    def setProfiling(self, enabled):
        self._check(self._L.sdfr_set_profiling(self._h, 1 if enabled else 0))

    def setStream(self, stream_handle):
        self._check(self._L.sdfr_set_stream(self._h, ctypes.c_void_p(stream_handle)))

    # bool render(FullscreenQuad&, GPUProfiler&, Camera&)
    def render(self, camera=None, width=1200, height=800, out=None, fmt=RGBA32F, pixel_stats=False):
        """Renders one frame.  `out` may be a CUDA/HIP torch tensor ([H,W,4] float32 or
        float16) -> device-to-device, asynchronous on the renderer's stream; otherwise a host
        numpy array is returned (synchronous).  pixel_stats: True (host path) also returns
        [H,W,3] uint32 {rays, march evaluations, hits}; with a device `out` it may be a device
        int32/uint32 tensor [H,W,3] that receives them."""
        if camera is not None:
            self.setCamera(camera)
        if out is not None and hasattr(out, "data_ptr"):
            assert out.is_cuda and out.is_contiguous() and out.numel() == width * height * 4
            pst = None
            if pixel_stats is not False and pixel_stats is not None:
                assert hasattr(pixel_stats, "data_ptr") and pixel_stats.is_cuda and pixel_stats.is_contiguous()
                assert pixel_stats.numel() == width * height * 3 and pixel_stats.element_size() == 4
                pst = ctypes.c_void_p(pixel_stats.data_ptr())
            self._check(self._L.sdfr_render(self._h, width, height, ctypes.c_void_p(out.data_ptr()), fmt, 0, pst))
            return out
        dt = np.float32 if fmt == RGBA32F else np.float16
        img = np.zeros((height, width, 4), dt) if out is None else out
        st = np.zeros((height, width, 3), np.uint32) if pixel_stats else None
        self._check(self._L.sdfr_render(self._h, width, height, img.ctypes.data_as(ctypes.c_void_p), fmt, 1,
                                        st.ctypes.data_as(ctypes.c_void_p) if pixel_stats else None))
        return (img, st) if pixel_stats else img

    def registerHostTarget(self, array):
        """Page-lock a numpy image that render(out=array) will fill every frame (sdfr_register_host_target); None
        unregisters.  The array must stay alive and unmoved while registered."""
        if array is None:
            self._check(self._L.sdfr_register_host_target(self._h, None, 0))
            self._host_target = None
        else:
            assert array.flags["C_CONTIGUOUS"]
            self._check(self._L.sdfr_register_host_target(self._h, array.ctypes.data_as(ctypes.c_void_p), array.nbytes))
            self._host_target = array  # keeps it alive

    def setStripSplit(self, priv_count, priv_period):
        """Of every priv_period strips the first priv_count are private to the root (renderPrivateStrips),
        the others are shared round-robin (renderStrips / assembleStrips).  (0, 1) = plain round-robin."""
        self._check(self._L.sdfr_set_strip_split(self._h, int(priv_count), int(priv_period)))
        self._split = (int(priv_count), int(priv_period))

    def renderPrivateStrips(self, width, height, out, fmt=RGBA32F):
        """Root only: render the private strips straight into the full-size device image `out`."""
        assert out.is_cuda and out.is_contiguous() and out.numel() == width * height * 4
        self._check(self._L.sdfr_render_private_strips(self._h, width, height, ctypes.c_void_p(out.data_ptr()), fmt))
        return out

    def renderStrips(self, width, height, rank, world, out, fmt=RGBA32F):
        """Multi-GPU: render this rank's 8-row strips into the compact device tensor `out`."""
        assert out.is_cuda and out.is_contiguous()
        assert out.numel() * out.element_size() == strip_buffer_bytes(width, height, world, fmt, getattr(self, "_split", (0, 1)))
        self._check(self._L.sdfr_render_strips(self._h, width, height, rank, world, ctypes.c_void_p(out.data_ptr()), fmt))
        return out

    def assembleStrips(self, width, height, world, gathered, out, fmt=RGBA32F):
        self._check(self._L.sdfr_assemble_strips(self._h, width, height, world, ctypes.c_void_p(gathered.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr()), fmt))
        return out

    def renderGather(self, comm, width, height, out=None, fmt=RGBA32F, wire=None):
        """Multi-GPU frame through the library's own RCCL gather (sdfr_render_gather): every rank renders
        its strips, rank 0 receives them and assembles `out` (a device tensor of the full frame; None
        on the other ranks).  wire: strip format on the links (default: the packed one that matches fmt)."""
        if wire is None:
            wire = STRIP_RGB32F_A8 if fmt == RGBA32F else STRIP_RGB16F_A8
        ptr = None
        if out is not None:
            assert out.is_cuda and out.is_contiguous() and out.numel() == width * height * 4
            assert out.element_size() == (4 if fmt == RGBA32F else 2)
            ptr = ctypes.c_void_p(out.data_ptr())
        self._check(self._L.sdfr_render_gather(self._h, comm._c, width, height, ptr, fmt, wire))
        return out

    def getTimings(self):
        """GPUProfiler::getResults: {name: ms} of the last render and the last postprocess, in frame order."""
        buf = (_CTiming * 48)()
        n = self._L.sdfr_get_timings(self._h, buf, 48)
        if n < 0:
            self._check(n)
        return {buf[i].name.decode(): buf[i].ms for i in range(min(n, 48))}

    def postprocess(self, scene16, bloom_scratch, out8):
        """HDR::process on device tensors: scene16/bloom_scratch [H,W,4] float16, out8 [H,W,4] uint8."""
        H, W = scene16.shape[0], scene16.shape[1]
        assert scene16.is_cuda and scene16.is_contiguous() and bloom_scratch.is_contiguous() and out8.is_contiguous()
        assert bloom_scratch.numel() == scene16.numel() == out8.numel()
        self._check(self._L.sdfr_postprocess(self._h, W, H, ctypes.c_void_p(scene16.data_ptr()), ctypes.c_void_p(bloom_scratch.data_ptr()),
                                             ctypes.c_void_p(out8.data_ptr())))
        return out8

    def selftestMath(self, what, constant=1.0):
        """Exhaustive GPU check of the fast exact sqrt (what=0) / constant division (what=1); returns mismatches."""
        n = ctypes.c_uint64()
        self._check(self._L.sdfr_selftest_math(self._h, int(what), float(constant), ctypes.byref(n)))
        return int(n.value)

    def sync(self):
        self._check(self._L.sdfr_sync(self._h))

    def setFramesInFlight(self, n):
        """sdfr_set_frames_in_flight: 2 = render() alternates between two internal streams and workspaces, so that a frame starts
        while the one before drains (render into two images in turn; sync() waits for both); 1 = the default."""
        self._check(self._L.sdfr_set_frames_in_flight(self._h, int(n)))

    def waitFrame(self, stream=None):
        """sdfr_wait_frame: `stream` (a raw hipStream_t / torch stream's .cuda_stream; None = the default stream) waits on the
        device for the frame submitted last"""
        self._check(self._L.sdfr_wait_frame(self._h, ctypes.c_void_p(stream)))

    def getStats(self):
        s = Stats()
        self._check(self._L.sdfr_get_stats(self._h, ctypes.byref(s)))
        return s


class Comm:
    """One RCCL communicator per process and GPU (sdfr_comm_*).  Rank 0 makes the id with
    Comm.unique_id() and hands it to the other ranks by any means (bench.py: torch.distributed
    broadcast); creation is collective."""

    @staticmethod
    def _one_rccl_per_process():
        """PyTorch ships a librccl of its own (torch/lib/librccl.so) and loads it by PATH, whatever the process holds
        already.  libsdfr.so opens a copy the process maps before any other (sdfr_comm.cpp), so the process runs one RCCL
        if torch's is there FIRST; a process that made a communicator and imported torch afterwards would run two.
        load_library() imports torch already (one HIP runtime); this keeps the property where someone loads the
        library by other means."""
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass

    @staticmethod
    def library_info():
        """(path of the librccl serving libsdfr.so, ncclGetVersion code, distinct librccl files mapped by the process)"""
        Comm._one_rccl_per_process()
        L = load_library()
        path = ctypes.create_string_buffer(1024)
        version, copies = ctypes.c_int(0), ctypes.c_int(0)
        rc = L.sdfr_comm_library_info(path, len(path), ctypes.byref(version), ctypes.byref(copies))
        if rc != SDFR_OK:
            raise SdfrError(rc, L.sdfr_comm_last_error(None).decode())
        return path.value.decode(), version.value, copies.value

    @staticmethod
    def unique_id():
        Comm._one_rccl_per_process()
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        L = load_library()
        rc = L.sdfr_comm_unique_id(buf)
        if rc != SDFR_OK:
            raise SdfrError(rc, L.sdfr_comm_last_error(None).decode())
        return bytes(buf.raw)

    def __init__(self, uid, rank, world, device=0):
        Comm._one_rccl_per_process()
        self._L = load_library()
        self._c = ctypes.c_void_p()
        assert len(uid) == COMM_ID_BYTES
        rc = self._L.sdfr_comm_create(ctypes.c_char_p(uid), int(rank), int(world), int(device), ctypes.byref(self._c))
        if rc != SDFR_OK:
            raise SdfrError(rc, self._L.sdfr_comm_last_error(None).decode())
        self.rank, self.world, self.device = int(rank), int(world), int(device)

    def selftest(self, nbytes=1 << 20, stream_handle=None):
        """Ring exchange of nbytes over the communicator, compared at the destination (blocking)."""
        rc = self._L.sdfr_comm_selftest(self._c, int(nbytes), ctypes.c_void_p(stream_handle))
        if rc != SDFR_OK:
            raise SdfrError(rc, self._L.sdfr_comm_last_error(self._c).decode())
        return True

    def close(self):
        """sdfr_comm_close: drains the streams the communicator's transfers ran on, then ncclCommFinalize + ncclCommDestroy,
        bounded in time (SDFR_COMM_CLOSE_TIMEOUT_S, default 30 s): a teardown that does not finish raises SdfrError with the
        call it was stuck in instead of holding the process.  Collective in effect -- every rank closes.  Not called
        implicitly (a communicator still open when the process ends is reclaimed with it)."""
        if self._c:
            c, self._c = self._c, ctypes.c_void_p()
            rc = self._L.sdfr_comm_close(c)
            if rc != SDFR_OK:
                raise SdfrError(rc, self._L.sdfr_comm_last_error(None).decode())


class HDR:
    """The post-processing stage that consumes the render target.  Mirrors the reference's
    `class HDR` (Engine/Postprocessing.h:12-43): init / getRenderTarget / process.  The three
    RGBA16F textures of the reference become two device tensors (render target + one bloom
    buffer; the second bloom texture is never materialised, see sdfr_post.hip)."""

    def __init__(self, renderer):
        self._r = renderer
        self.width = self.height = 0

    def init(self, width, height):
        import torch

        self.width, self.height = int(width), int(height)
        self._target = torch.zeros((self.height, self.width, 4), dtype=torch.float16, device="cuda")
        self._bloom = torch.empty_like(self._target)
        self._ldr = torch.empty((self.height, self.width, 4), dtype=torch.uint8, device="cuda")
        return True

    def getRenderTarget(self):
        """The RGBA16F tensor SDFRenderer.render(..., out=..., fmt=RGBA16F) draws into."""
        return self._target

    def process(self, scene=None):
        """bloom + tone map of the render target (or of `scene`, an [H,W,4] float16 cuda
        tensor) -> [H,W,4] uint8 cuda tensor (R8G8B8A8_UNORM)."""
        src = self._target if scene is None else scene
        return self._r.postprocess(src, self._bloom, self._ldr)


def _shared_index(strip, split):
    """Index of a frame strip among the shared strips, or None if it is private (split = (m, M))."""
    m, M = split
    if m == 0:
        return strip
    j = strip % M
    return None if j < m else (strip // M) * (M - m) + (j - m)


def assemble_strips_host(width, height, world, gathered, split=(0, 1), image=None):
    """Host (numpy) statement of the strip layout: gathered[world, strip_buffer_pixels, C] ->
    image[height, width, C] (private strips of `split` are left as they are in `image`).  Used by
    the CPU tests of the multi-rank path; the GPU path is sdfr_assemble_strips."""
    gathered = np.asarray(gathered)
    C = gathered.shape[-1]
    per_rank_rows = strip_buffer_pixels_host(width, height, world, split) // width
    g = gathered.reshape(world, per_rank_rows, width, C)
    img = np.zeros((height, width, C), gathered.dtype) if image is None else image
    for py in range(height):
        t = _shared_index(py // STRIP_ROWS, split)
        if t is not None:
            img[py] = g[t % world, (t // world) * STRIP_ROWS + py % STRIP_ROWS]
    return img


def pack_strip_host(compact_rgba):
    """Host statement of SDFR_STRIP_RGB32F_A8: [n, 4] float32 (alpha 0 or 1) -> uint8 buffer of
    (13 n + 3) // 4 * 4 bytes: n rgb float triples, then n flag bytes."""
    c = np.ascontiguousarray(compact_rgba, np.float32).reshape(-1, 4)
    n = c.shape[0]
    out = np.zeros(((13 * n + 3) // 4 * 4,), np.uint8)
    out[:12 * n] = np.ascontiguousarray(c[:, :3]).view(np.uint8).reshape(-1)
    out[12 * n:13 * n] = (c[:, 3] != 0).astype(np.uint8)
    return out


def unpack_strip_host(packed, n):
    """Inverse of pack_strip_host: -> [n, 4] float32."""
    packed = np.asarray(packed, np.uint8)
    c = np.zeros((n, 4), np.float32)
    c[:, :3] = packed[:12 * n].view(np.float32).reshape(n, 3)
    c[:, 3] = packed[12 * n:13 * n].astype(np.float32)
    return c


def pack_strip16_host(compact_rgba):
    """Host statement of SDFR_STRIP_RGB16F_A8: [n, 4] float32 (alpha 0 or 1) -> uint8 buffer of
    (7 n + 3) // 4 * 4 bytes: n rgb half triples (round to nearest even), then n flag bytes."""
    c = np.ascontiguousarray(compact_rgba, np.float32).reshape(-1, 4)
    n = c.shape[0]
    out = np.zeros(((7 * n + 3) // 4 * 4,), np.uint8)
    with np.errstate(over="ignore"):
        out[:6 * n] = np.ascontiguousarray(c[:, :3].astype(np.float16)).view(np.uint8).reshape(-1)
    out[6 * n:7 * n] = (c[:, 3] != 0).astype(np.uint8)
    return out


def unpack_strip16_host(packed, n):
    """Inverse of pack_strip16_host: -> [n, 4] float16 (the RGBA16F pixels)."""
    packed = np.asarray(packed, np.uint8)
    c = np.zeros((n, 4), np.float16)
    c[:, :3] = packed[:6 * n].view(np.float16).reshape(n, 3)
    c[:, 3] = packed[6 * n:7 * n].astype(np.float16)
    return c


def strip_buffer_pixels_host(width, height, world, split=(0, 1)):
    strips = (height + STRIP_ROWS - 1) // STRIP_ROWS
    shared = sum(1 for s in range(strips) if _shared_index(s, split) is not None)
    return ((shared + world - 1) // world) * STRIP_ROWS * width


def private_rows_host(height, split):
    """Global rows of the private strips (rendered by the root straight into the image)."""
    return [r for r in range(height) if _shared_index(r // STRIP_ROWS, split) is None]


def strip_rows_of_rank(height, rank, world, split=(0, 1)):
    """Global rows of the shared strips owned by `rank`, in the order they appear in its compact buffer."""
    rows = []
    strips = (height + STRIP_ROWS - 1) // STRIP_ROWS
    for s in range(strips):
        t = _shared_index(s, split)
        if t is None or t % world != rank:
            continue
        for k in range(STRIP_ROWS):
            if s * STRIP_ROWS + k < height:
                rows.append(s * STRIP_ROWS + k)
    return rows
