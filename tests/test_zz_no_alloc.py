"""rt_render_device_noalloc (workspace in the library's static device array, no hipMalloc) gives the
frames rt_render_device gives: two calls in a row, the second on a smaller input, through the C ABI;
it leaves the device's free memory as it found it, and its launch sequence can be captured into a HIP
graph and replayed."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def hip_runtime():
    return ctypes.CDLL("libamdhip64.so")


def free_bytes(hip):
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def frames(ds, dim, cam, cfg):
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    want = torch.zeros((dim.height, dim.width, 3), dtype=torch.uint8, device=dev)
    got = torch.full((dim.height, dim.width, 3), 7, dtype=torch.uint8, device=dev)
    ds.render_device(dim, cam, cfg, want.data_ptr(), stream)
    assert ds.frame_dropped() == 0
    ds.render_device_noalloc(dim, cam, cfg, got.data_ptr(), stream)
    torch.cuda.synchronize()                 # (the call records no event: rt_frame_status relies on the caller's synchronisation)
    assert ds.frame_dropped() == 0
    return got.cpu().numpy(), want.cpu().numpy()


@pytest.mark.parametrize("name", ["glass_monkey", "many_balls", "subtraction_outside1"])
def test_noalloc_entry_point_matches_rt_render_device(name, render_mod, scene_cache):
    s = scene_cache(name)
    ds = render_mod.DeviceScene(s, device=0)
    # first call: the largest frame the entry point takes (256 x 128 = RT_NOALLOC_MAX_PIXELS)
    dim, cam, cfg = s.view(width=256, height=128, anti_alias=True, seed=1)
    got, want = frames(ds, dim, cam, cfg)
    assert want.max() > 0 and np.array_equal(got, want), name
    # second call, smaller and ragged: nothing of the first frame may show through
    dim, cam, cfg = s.view(width=97, height=61, anti_alias=True, seed=2)
    got, want = frames(ds, dim, cam, cfg)
    assert np.array_equal(got, want), name
    ds.close()


def test_noalloc_limits_are_reported(render_mod, scene_cache):
    s = scene_cache("glass_monkey")
    ds = render_mod.DeviceScene(s, device=0)
    buf = torch.zeros((300, 300, 3), dtype=torch.uint8, device="cuda:0")
    dim, cam, cfg = s.view(width=300, height=300)
    with pytest.raises(render_mod.RtError) as e:
        ds.render_device_noalloc(dim, cam, cfg, buf.data_ptr())
    assert e.value.code == 5 and "RT_NOALLOC_MAX_PIXELS" in str(e.value)
    ds.close()
    a = scene_cache("light_room")                       # square light: persistent-kernel schedule only
    da = render_mod.DeviceScene(a, device=0)
    dim, cam, cfg = a.view(width=64, height=64)
    with pytest.raises(render_mod.RtError) as e:
        da.render_device_noalloc(dim, cam, cfg, buf.data_ptr())
    assert e.value.code == 2
    da.close()


def test_noalloc_entry_point_allocates_nothing(render_mod, scene_cache):
    """hipMemGetInfo's free bytes are the same before and after the calls (rt_render_device on a fresh scene, by
    contrast, takes its workspace from hipMalloc)."""
    hip = hip_runtime()
    s = scene_cache("glass_monkey")
    ds = render_mod.DeviceScene(s, device=0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    dim, cam, cfg = s.view(width=256, height=128, anti_alias=True, seed=1)
    out = torch.zeros((dim.height, dim.width, 3), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    before = free_bytes(hip)
    for _ in range(3):
        ds.render_device_noalloc(dim, cam, cfg, out.data_ptr(), stream)
    torch.cuda.synchronize()
    assert ds.frame_dropped() == 0
    assert free_bytes(hip) == before
    ds.render_device(dim, cam, cfg, out.data_ptr(), stream)     # the allocating entry point, for contrast
    torch.cuda.synchronize()
    assert free_bytes(hip) < before
    ds.close()


def test_noalloc_entry_point_under_stream_capture(render_mod, scene_cache):
    """The call is made of stream operations only: captured into a HIP graph (global capture mode, which refuses
    allocations and synchronisations) and replayed, it renders the frame rt_render_device renders."""
    hip = hip_runtime()
    s = scene_cache("glass_monkey")
    ds = render_mod.DeviceScene(s, device=0)
    dev = torch.device("cuda", 0)
    dim, cam, cfg = s.view(width=160, height=96, anti_alias=True, seed=3)
    want = torch.zeros((dim.height, dim.width, 3), dtype=torch.uint8, device=dev)
    got = torch.zeros_like(want)
    ds.render_device(dim, cam, cfg, want.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
    ds.render_device_noalloc(dim, cam, cfg, got.data_ptr(), stream.value)   # once outside the capture: kernels loaded, LDS limits granted
    assert hip.hipStreamSynchronize(stream) == 0
    got.zero_()
    torch.cuda.synchronize()
    graph, graph_exec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0                          # hipStreamCaptureModeGlobal
    try:
        ds.render_device_noalloc(dim, cam, cfg, got.data_ptr(), stream.value)
    finally:
        rc = hip.hipStreamEndCapture(stream, ctypes.byref(graph))
    assert rc == 0 and graph.value
    assert hip.hipGraphInstantiate(ctypes.byref(graph_exec), graph, None, None, 0) == 0
    for _ in range(2):                                                         # replayed twice: the frame leaves nothing behind that the next one trips over
        got.zero_()
        torch.cuda.synchronize()
        assert hip.hipGraphLaunch(graph_exec, stream) == 0
        assert hip.hipStreamSynchronize(stream) == 0
        assert torch.equal(got, want)
    assert ds.frame_dropped() == 0
    hip.hipGraphExecDestroy(graph_exec)
    hip.hipGraphDestroy(graph)
    hip.hipStreamDestroy(stream)
    ds.close()
