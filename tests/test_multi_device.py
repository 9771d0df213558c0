"""The multi-device half of the C ABI: rt_partition_spp, rt_multi_create / upload / render.
CPU: the partition arithmetic (every sample index rendered exactly once, remainder to the lowest parts, identical
to raytracing_amd.dist.partition_spp that the one-process-per-GPU path uses) and argument errors that need no
device.  GPU (this pool's boxes have ONE GPU): n = 1 through the same code path as n = 8 -- ncclCommInitAll, the
spp share with spp_offset, ncclReduce in place on the root, the mean -- against rt_render; and n = 2, 3, 5 "devices"
that are all GPU 0 (rt_debug_multi lets an ordinal be listed several times; the exchange step is then the copy-and-add
fallback, since RCCL refuses two ranks on one device): the per-device shares, the more-devices-than-samples branch,
shares that need several launches, the root-only mean."""
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


@pytest.mark.gpu
def test_multi_render_several_entries_of_one_gpu(ctx, api):
    """The n > 1 loop of rt_multi_render on a one-GPU box (include/rtx_hip_debug.h rt_debug_multi).  Same image as
    rt_render up to the fp32 rounding of the per-part sums (each part's u64 sums are narrowed to fp32 before they are
    added: 2e-7 relative to a radiance of ~1), same ray count exactly."""
    api.debug_multi(1)
    try:
        cases = (  # (devices, scene, W, H, spp, spp_offset)
            ([0, 0], "Random", 200, 120, 33, 0),          # odd spp: shares 17 + 16
            ([0, 0, 0], "FinalStatic", 97, 61, 16, 5),    # 6 + 5 + 5, ragged image, an offset
            ([0, 0, 0, 0, 0], "PerlinMarbled", 64, 48, 3, 0),  # more devices than samples: two of them contribute zeros
            ([0, 0], "Balls", 2048, 2048, 1040, 0),       # 520 spp per share = 2.18e9 samples: two launches per share
        )
        for devices, name, W, H, spp, off in cases:
            m = api.MultiContext(devices)
            try:
                w = api.World(name, seed=1, aspect=W / H)
                depth = 8 if W >= 2048 else 100
                p = api.make_params(W, H, spp, max_depth=depth, seed=w.render_seed, spp_offset=off)
                m.upload(w.scene)
                ctx.upload(w.scene)
                img, st = m.render(w.camera, p)
                ref, s1 = ctx.render(w.camera, p)
                assert st.n_devices == len(devices) and st.exchange == 1 and st.samples == W * H * spp
                assert st.rays_traced == s1.rays_traced
                err = float(np.abs(img.astype(np.float64) - ref).max())
                print(name, devices, "max abs diff", err, "kernel ms per entry", [round(st.device_kernel_ms[i], 2) for i in range(len(devices))],
                      "exchange ms", round(st.reduce_ms, 3))
                assert err < 2e-7 * max(1.0, float(ref.max())) * len(devices)
                img2, st2 = m.render(w.camera, p)  # again on the same rt_multi: buffers reused
                assert np.array_equal(img, img2) and st2.rays_traced == st.rays_traced
                with pytest.raises(api.RtError):  # an argument error leaves nothing in flight and the object usable
                    m.render(w.camera, api.make_params(1, 1, 1))
                img3, _ = m.render(w.camera, api.make_params(W // 2 if W > 200 else W, H // 2 if W > 200 else H, 2, max_depth=depth, seed=1))
                assert np.isfinite(img3).all()
            finally:
                m.close()
        # the same fallback with ONE entry and RCCL switched off: no exchange step at all
        api.debug_multi(2)
        m = api.MultiContext([0])
        try:
            w = api.World("Random", seed=1, aspect=1.5)
            p = api.make_params(120, 80, 9, seed=w.render_seed)
            m.upload(w.scene)
            ctx.upload(w.scene)
            img, st = m.render(w.camera, p)
            ref, s1 = ctx.render(w.camera, p)
            assert st.exchange == 1 and st.rays_traced == s1.rays_traced and np.abs(img.astype(np.float64) - ref).max() < 2e-7
        finally:
            m.close()
    finally:
        api.debug_multi(0)
    with pytest.raises(api.RtError):  # the default again: a repeated ordinal is refused
        api.MultiContext([0, 0])
