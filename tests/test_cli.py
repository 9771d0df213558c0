"""rtx_cli: the reference's command line (config.rs:25-69) over the C ABI.  CPU tests cover argument handling
(no GPU call is reached before the scene name and the NonZero checks); the GPU test checks the PPM against the
Python binding's render + encode of the same arguments."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "raytracing_amd", "rtx_cli")


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=300)


def test_cli_built():
    assert os.access(CLI, os.X_OK), "raytracing_amd/rtx_cli missing: run __graft_entry__.build()"


def test_cli_help_lists_reference_scenes():
    r = run("--help")
    assert r.returncode == 2
    for name in ["Random", "TwoSpheres", "Balls", "BirdsEyeView", "ValueNoise1", "ValueNoise2", "ValueNoisePink",
                 "ValueTurbulence", "ValueMarbled", "PerlinNoise1", "PerlinNoise2", "PerlinNoisePink",
                 "PerlinTurbulence", "PerlinMarbled"]:  # config.rs:81-99
        assert name in r.stderr


def test_cli_unknown_scene_message():
    r = run("random")  # names are exact-case (config.rs:71-79)
    assert r.returncode == 2
    assert "\"random\" isn't a Scene." in r.stderr


@pytest.mark.parametrize("args", [("-s", "0"), ("-s", "65536"), ("-w", "0"), ("-h", "0"), ("-m", "0"),
                                  ("-w", "abc"), ("--bogus",)])
def test_cli_rejects_bad_values(args):
    r = run(*args)
    assert r.returncode == 2 and "error:" in r.stderr


@pytest.mark.gpu
def test_cli_ppm_matches_binding(api, ctx, tmp_path):
    out = tmp_path / "o.ppm"
    r = run("-w", "96", "-h", "64", "-s", "8", "-m", "20", "-r", "5", "-o", str(out), "TwoSpheres")
    assert r.returncode == 0, r.stderr
    assert "Rending time elapsed" in r.stderr
    raw = out.read_bytes()
    head = b"P6\n96 64\n255\n"
    assert raw.startswith(head)
    pix = np.frombuffer(raw[len(head):], dtype=np.uint8).reshape(64, 96, 3)
    w = api.World("TwoSpheres", seed=5, aspect=96 / 64)
    ctx.upload(w.scene)
    p = api.make_params(96, 64, 8, max_depth=20, seed=w.render_seed)
    img, _ = ctx.render(w.camera, p)
    enc, bad = ctx.encode(img.reshape(-1, 3))
    assert bad == 0
    ref = np.stack([(enc >> 16) & 255, (enc >> 8) & 255, enc & 255], -1).astype(np.uint8).reshape(64, 96, 3)
    assert np.array_equal(pix, ref)


@pytest.mark.gpu
def test_cli_gpus_flag_renders_through_rt_multi(tmp_path):
    """--gpus n goes through rt_multi_* (RCCL communicator, spp split, reduce); with n = 1 the PPM is the one-device
    PPM (the two means differ by one fp32 rounding at most: allow one grey level in a handful of pixels)."""
    a, b = tmp_path / "a.ppm", tmp_path / "b.ppm"
    args = ("-w", "96", "-h", "64", "-s", "8", "-m", "20", "-r", "5", "Balls")
    r1 = run(*args, "-o", str(a))
    r2 = run(*args, "-g", "1", "-o", str(b))
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    assert "1 device(s), RCCL reduce" in r2.stderr
    pa = np.frombuffer(a.read_bytes(), dtype=np.uint8).astype(int)
    pb = np.frombuffer(b.read_bytes(), dtype=np.uint8).astype(int)
    assert pa.shape == pb.shape and np.abs(pa - pb).max() <= 1 and (pa != pb).sum() <= 8
    assert run(*args, "-g", "0").returncode == 2 and run(*args, "-g", "99").returncode == 2
