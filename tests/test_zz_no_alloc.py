"""rt_render_accum_device_noalloc (the entry point for callers that cannot allocate: its workspace is a static
device array inside the library) against rt_render_accum_device, through the C ABI.  The two share their kernels
and the radiance sums are integers, so the outputs must be BIT-IDENTICAL (the bar of the other image-level
checks of identical sample sets).  Two calls in a row, the second with a smaller image: the shared array keeps
the first call's contents and the second must not see them."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(ctx, api, torch, w, W, H, spp, **kw):
    p = api.make_params(W, H, spp, seed=w.render_seed, **kw)
    a = torch.full((H, W, 3), -1.0, dtype=torch.float32, device="cuda:0")
    b = torch.full((H, W, 3), -2.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()  # the library renders on its own stream
    sa = ctx.render_accum_device(w.camera, p, a.data_ptr())
    sb = ctx.render_accum_device_noalloc(w.camera, p, b.data_ptr())
    a, b = a.cpu().numpy(), b.cpu().numpy()
    assert np.array_equal(a, b), float(np.abs(a - b).max())
    assert sa.rays_traced == sb.rays_traced and sa.samples == sb.samples == W * H * spp
    assert sb.n_launches == sa.n_launches
    assert np.isfinite(b).all() and b.min() >= 0 and b.max() <= spp and b.sum() > 0
    return b


def test_noalloc_entry_allocates_no_device_memory(api):
    """On a FRESH context, before the allocating twin has ever run on it (so nothing of the render has been allocated
    or warmed up on its behalf): the device's free memory (hipMemGetInfo through torch.cuda.mem_get_info) is the same
    after the call as before it, while the allocating twin then takes W*H*3 u64 accumulators from it."""
    import torch
    ctx = api.Context(0)
    try:
        w = api.World("FinalStatic", seed=1, aspect=640 / 400)
        ctx.upload(w.scene)
        W, H = 640, 400
        p = api.make_params(W, H, 4, seed=w.render_seed)
        out = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        st = ctx.render_accum_device_noalloc(w.camera, p, out.data_ptr())
        free1, _ = torch.cuda.mem_get_info(0)
        assert st.samples == W * H * 4 and out.sum().item() > 0
        assert free1 == free0, (free0, free1)
        ref = torch.zeros_like(out)
        torch.cuda.synchronize()
        free2, _ = torch.cuda.mem_get_info(0)
        ctx.render_accum_device(w.camera, p, ref.data_ptr())
        free3, _ = torch.cuda.mem_get_info(0)
        assert torch.equal(out, ref)
        assert free2 - free3 >= W * H * 3 * 8 // 2, (free2, free3)  # (the allocator rounds: at least half of 6.1 MB shows)
    finally:
        ctx.close()


def test_noalloc_entry_matches_the_allocating_one(api):
    import torch
    ctx = api.Context(0)  # a context of its own: passes alone in a fresh process and after the other tests
    try:
        w = api.World("Random", seed=1, aspect=304 / 200)  # moving spheres, checker, lens, shutter
        ctx.upload(w.scene)
        big = _both(ctx, api, torch, w, 304, 200, 24)
        w2 = api.World("Random", seed=1, aspect=75 / 41)   # smaller, ragged image right after: stale sums must not leak
        small = _both(ctx, api, torch, w2, 75, 41, 9, spp_offset=5)
        assert small.shape == (41, 75, 3) and big.shape == (200, 304, 3)
        # a noise scene (the other kernel variant) and the largest image the workspace holds
        w3 = api.World("PerlinMarbled", seed=1, aspect=1024 / 1024)
        ctx.upload(w3.scene)
        _both(ctx, api, torch, w3, 1024, 1024, 2)
        # beyond the limit: a clear error, nothing rendered
        w4 = api.World("PerlinMarbled", seed=1, aspect=1200 / 900)
        out = torch.zeros(900, 1200, 3, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        with pytest.raises(api.RtError) as e:
            ctx.render_accum_device_noalloc(w4.camera, api.make_params(1200, 900, 1), out.data_ptr())
        assert e.value.code == api.RT_ERR_BAD_ARG and "RT_NOALLOC_MAX_PIXELS" in str(e.value)
        assert not out.any().item()
    finally:
        ctx.close()
