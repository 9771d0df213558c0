"""The multi-device half of the C ABI: rt_partition_spp, rt_multi_create / upload / render.
CPU: the partition arithmetic (every sample index rendered exactly once, remainder to the lowest parts, identical
to raytracing_amd.dist.partition_spp that the one-process-per-GPU path uses) and argument errors that need no
device.  GPU (this pool's boxes have ONE GPU): n = 1 through the same code path as n = 8 -- ncclCommInitAll, the
spp share with spp_offset, ncclReduce in place on the root, the mean -- against rt_render."""
import numpy as np
import pytest

from raytracing_amd import dist as rdist


@pytest.mark.parametrize("total", [0, 1, 7, 8, 500, 1000, 4096, 4097, 2**32 - 1])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 8, 16])
def test_partition_covers_every_sample_once(api, total, n):
    nxt = 0
    for part in range(n):
        first, count = api.partition_spp(total, n, part)
        assert (first, count) == rdist.partition_spp(total, n, part)
        assert first == nxt and count in (total // n, total // n + 1)
        nxt = first + count
    assert nxt == total
    counts = [api.partition_spp(total, n, k)[1] for k in range(n)]
    assert counts == sorted(counts, reverse=True)  # the remainder goes to the lowest parts


def test_partition_and_create_argument_errors(api):
    for bad in ((10, 0, 0), (10, 4, 4), (10, 4, 9)):
        with pytest.raises(api.RtError) as e:
            api.partition_spp(*bad)
        assert e.value.code == api.RT_ERR_BAD_ARG
    for devices in ([], [0, 0], list(range(17))):
        with pytest.raises(api.RtError) as e:
            api.MultiContext(devices)
        assert e.value.code == api.RT_ERR_BAD_ARG


@pytest.mark.gpu
def test_multi_render_one_device_matches_rt_render(ctx, api):
    m = api.MultiContext([0])
    try:
        for name, W, H, spp, off in (("Random", 200, 120, 33, 0), ("PerlinMarbled", 97, 61, 16, 5)):
            w = api.World(name, seed=1, aspect=W / H)
            p = api.make_params(W, H, spp, seed=w.render_seed, spp_offset=off)
            m.upload(w.scene)
            ctx.upload(w.scene)
            img, st = m.render(w.camera, p)
            ref, s1 = ctx.render(w.camera, p)
            assert st.n_devices == 1 and st.samples == W * H * spp and st.rays_traced == s1.rays_traced
            assert st.kernel_ms > 0 and st.reduce_ms >= 0 and st.device_kernel_ms[0] == st.kernel_ms
            # same integer sums; the mean is taken in fp32 here and in f64 there: one rounding apart
            assert np.abs(img.astype(np.float64) - ref).max() < 2e-7
        with pytest.raises(api.RtError) as e:
            m.render(w.camera, api.make_params(1, 1, 1))
        assert e.value.code == api.RT_ERR_BAD_ARG
    finally:
        m.close()
    with pytest.raises(api.RtError) as e:
        api.MultiContext([0, 99])  # no such device on this box
    assert e.value.code in (api.RT_ERR_BAD_ARG, api.RT_ERR_NO_DEVICE)
