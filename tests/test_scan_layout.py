"""The device-free layout of an rt_scene for the scan (raytracing_amd/csrc/scan_layout.h: big / small split, scan
frame, planar list, Morton order, padding, widened scan-side radii, bound levels over a member's whole motion) on the
CPU, under AddressSanitizer + UBSan: tools/scan_layout_test.cpp checks on constructed and random scenes that every
sphere is laid out exactly once, that every member at time 0 and 1 lies inside its bound at every level in the rounded
fp32 numbers the kernel reads, that pads are never nominated, that planar members and bounds are in-plane to the bit,
that the split and the order do not depend on the scale of the scene, and the limits (0, 1, 4, 5, 16384 spheres,
all-moving, more big candidates than the f64 list holds, coincident duplicates)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scan_layout_invariants_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "raytracing_amd", "csrc"), "layout_test"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(ROOT, "raytracing_amd", "scan_layout_test"), "120"], capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "scan_layout_test: ok" in out.stdout and "runtime error" not in out.stderr
