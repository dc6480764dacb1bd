"""rt_render_device_noalloc (workspace in the library's static device array, no hipMalloc) gives the
frames rt_render_device gives: two calls in a row, the second on a smaller input, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def frames(ds, dim, cam, cfg):
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    want = torch.zeros((dim.height, dim.width, 3), dtype=torch.uint8, device=dev)
    got = torch.full((dim.height, dim.width, 3), 7, dtype=torch.uint8, device=dev)
    ds.render_device(dim, cam, cfg, want.data_ptr(), stream)
    assert ds.frame_dropped() == 0
    ds.render_device_noalloc(dim, cam, cfg, got.data_ptr(), stream)
    assert ds.frame_dropped() == 0
    torch.cuda.synchronize()
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
