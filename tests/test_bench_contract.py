"""CPU: the parts of bench.py that decide what evidence the driver's record holds -- the HBM-traffic stamp
(profiles/traffic.json is only reported while its `isa_sha` equals the hash of the C2 kernel's machine code in the
built library), the single-call table's entries VERDICT r4 asked for, and the scalars compact_summary() puts into
`roofline` and the trailing `summary` object.  No device, no oracle: pure host code."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "libvips_amd", "lib", "libvipship.so")

pytestmark = pytest.mark.skipif(not os.path.exists(LIB), reason="libvipship.so is not built")


def test_traffic_stamp_matches_the_built_kernel():
    """A kernel edit that changes the C2 kernel's instructions without new PMC passes would turn `roofline.traffic`
    into null on the driver's box: the stamp must be the hash of the kernel in THIS tree's library."""
    table = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    # round 6: BASELINE config 2 runs reduce_fused_u8x4_mfma_x (no horizontal halo); its gate name is the longer key
    entry = table["reduce_fused_u8_mfma_x"]
    assert bench.kernel_isa_sha(entry["symbol"]) == entry["isa_sha"]
    assert bench.traffic_for("reduce_fused_u8_mfma_x") == entry["traffic_bytes"]
    assert entry["traffic_bytes"] / entry["algorithmic_bytes"] < 1.06  # (1.12 with the halos: VERDICT r5 item 3)
    # the counters it was stamped from are committed, and say what the entry says
    assert os.path.exists(os.path.join(ROOT, entry["profile"]))
    assert entry["traffic_bytes"] == entry["fetch_bytes_x2"] + entry["write_bytes"]
    assert 1.0 < entry["traffic_bytes"] / entry["algorithmic_bytes"] < 1.2


def test_isa_hash_is_of_machine_code_not_of_a_name():
    assert bench.kernel_isa_sha("no_such_kernel_in_the_library") is None
    a = bench.kernel_isa_sha("reduce_fused_u8x4_mfmaILi6ELi1ELi4ELb1ELi0ELb1ELi256ELi1EE")
    b = bench.kernel_isa_sha("reduce_fused_u8x4_mfmaILi7ELi1ELi4ELb1ELi0ELb1ELi256ELi1EE")
    assert a and b and a != b  # (two instantiations of one source: different instructions, different stamps)


def test_ops_table_has_the_calls_of_the_path():
    names = [op["name"] for op in bench.ops_table()]
    assert len(names) == len(set(names))
    for want in ("reducev_8", "reduceh_8", "reduce_rgb_8", "reduce_rgb_7.3", "resize_rgb_to_1000", "thumbnail_500",
                 "shrinkv_4", "shrinkh_4", "convi_3x3_u8", "convi_5x5_u8", "convi_3x3_u16", "gaussblur_s2_u8",
                 "gaussblur_s8_u8", "gaussblur_s2_u16", "gaussblur_s8_u16", "gaussblur_s2_f32",
                 "colourspace_srgb_lab_u8", "colourspace_srgb_labs_u8", "sharpen_u8", "reduce_rgba16_8",
                 # round 6: the rows VERDICT r5 found untimed
                 "reduce_f32_8", "shrink_f32_4", "shrinkv_rgba16_4", "shrinkh_rgba16_4", "unpremultiply_u8",
                 "thumbnail_rgba_500", "resize_bicubic_x2.5", "colourspace_lab_xyz_f32", "colourspace_xyz_scrgb_f32",
                 "conva_5x5_u8", "sharpen_u8_smooth"):
        assert want in names, want
    for op in bench.ops_table():
        assert ("chain" in op) != ("mask" in op), op["name"]  # every entry has exactly one reference recipe


def _synthetic_full_line(n_ops=60):
    """A line of the shape run_c2 + the configs + the single-call table make, with MORE single calls than the table
    holds and long free-text fields: what the final line must survive."""
    cpu = {"value": 2568.0, "unit": "Mpixels/s", "cores": 256, "kind": "reference", "threads_requested": 256,
           "cpus_allowed": 256, "sample": "full 16384x16384x4 u8 image, " * 20,
           "single_core": {"value": 317.6, "unit": "Mpixels/s", "cores": 1, "sample": "x" * 300}}
    return {
        "metric": "Mpixels/s, vips_reduce Lanczos3 16384x16384 uchar RGBA -> 2048x2048 (input pixels)",
        "value": 1385353.9, "unit": "Mpixels/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 0.1938,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic (LCG bytes, seed 12345 + rank, generated on device)",
        "config": {"workload": "vips_reduce(hshrink=8, vshrink=8, kernel=lanczos3) 16384x16384x4 u8 -> 2048x2048x4",
                   "images_per_step_per_gpu": 1, "partition": "one independent image per GPU"},
        "roofline": {"bound": "hbm", "kernel": "reduce_fused_u8_mfma", "achieved": 5605.0, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.69, "frac_cold": 0.69, "frac_settled": 0.71, "traffic": 123, "kernel_ms": 0.1945,
                     "algorithmic_bytes": 1090519040, "kernels": {"reduce_fused_u8_mfma": {"launches": 20, "mean_ms": 0.19}},
                     "measured_read_GBps": float("nan")},
        "settled": {"ms_per_step": 0.1913, "kernel_ms": 0.1905, "settle": {"launches": 120}},
        "parity": {"against": "oracle/_ref (compiled reference), whole output", "bit_exact": True, "checksum": 268743073531},
        "cpu_baseline": cpu,
        "configs": [
            {"name": "c3", "ms": 13.0, "frac_fp64": 0.36, "frac_hbm": 0.25, "float_input": {"ms": 14.3},
             "frac_of_fp64_stream": 0.48, "parity": {"bit_exact": True}, "cpu_baseline": cpu},
            {"name": "c4", "ms": 43.0, "frac": 0.6, "ms_per_image": 0.042, "images_per_gpu": 1024, "parity": {"bit_exact": True},
             "kernels_per_image": {"resize_stream_u8": {"gates_per_step": 16, "ms_per_image": 0.0428}}},
            {"name": "c5", "ms": 140.0, "tflops": 59.0, "frac": 0.75},
        ],
        "ops": [{"name": "op_number_%d_with_a_long_name" % i, "frac": 0.5, "ms": 0.1, "kernels": {"k": {"launches": 2}},
                 "parity": {"bit_exact": i != 7}, "cpu_baseline": cpu} for i in range(n_ops)],
    }


def _strict_loads(text):
    def reject(name):
        raise AssertionError("non-finite constant %s in the bench line" % name)
    return json.loads(text, parse_constant=reject)


def test_compact_summary_puts_scalars_where_the_record_keeps_them():
    line = _synthetic_full_line(2)
    bench.compact_summary(line)
    roof = line["roofline"]
    assert roof["c3_ms"] == 13.0 and roof["c3_ms_float_input"] == 14.3 and roof["c4_frac"] == 0.6 and roof["c5_tflops"] == 59.0
    assert roof["c4_resize_stream_u8_ms_per_image"] == 0.0428
    assert roof["op_op_number_0_with_a_long_name_frac"] == 0.5
    assert "others" not in roof and "summary" not in line  # (round 5's nested copies are gone)


def test_final_line_is_small_flat_and_strict_json():
    """BENCH_r05.json had `parsed: null` because the line was 24.5 KB.  The LAST stdout line is the
    compact record: under 6 000 characters whatever the table holds, strict JSON (no NaN / Infinity), the standard
    keys + a flat roofline + cpu_baseline + parity + frac_cold."""
    line = _synthetic_full_line(26)
    assert len(json.dumps(line)) > 15000  # the full table is what it was: too big for the driver
    bench.compact_summary(line)
    final = bench.final_line(line, "gpurun_out/bench_full.json")
    text = json.dumps(final, allow_nan=False)
    assert len(text) < 6000, len(text)
    back = _strict_loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "frac_cold"):
        assert key in back, key
    roof = back["roofline"]
    assert all(not isinstance(v, (dict, list)) for v in roof.values())
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_cold", "frac_settled", "c3_ms", "c4_frac",
                "c5_tflops", "op_op_number_3_with_a_long_name_frac"):
        assert key in roof, key
    assert roof["measured_read_GBps"] is None  # (a NaN became null, not a parse error)
    assert back["cpu_baseline"]["cores"] == 256 and back["cpu_baseline"]["kind"] == "reference"
    assert back["cpu_baseline"]["single_core_value"] == 317.6 and len(back["cpu_baseline"]["sample"]) <= 200
    assert back["parity"]["bit_exact"] is True and back["parity"]["entries_not_bit_exact"] == "op_number_7_with_a_long_name"
    assert back["parity"]["entries_unchecked"] == "c5"
    assert "configs" not in back and "ops" not in back and "settled" not in back
    assert back["full"] == "gpurun_out/bench_full.json"


def test_final_line_never_exceeds_the_limit():
    """More single calls than the limit has room for: the per-call fractions go (and the line says so) before the line
    grows past what the driver reads."""
    line = _synthetic_full_line(400)
    bench.compact_summary(line)
    final = bench.final_line(line, None)
    text = json.dumps(final, allow_nan=False)
    assert len(text) < 6000
    assert final["roofline"]["op_fracs_dropped"] == 400 and final["roofline"]["frac"] == 0.69
    _strict_loads(text)


def test_main_prints_the_compact_record_last(monkeypatch, capsys, tmp_path):
    """The order on stdout: the full table on an EARLIER line (prefixed: not a record), the compact record LAST."""
    line = _synthetic_full_line(26)
    bench.compact_summary(line)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.chdir(tmp_path)
    full_path = bench.write_full(line)
    assert full_path == os.path.join("gpurun_out", "bench_full.json")
    stored = json.load(open(os.path.join(str(tmp_path), full_path)))
    assert len(stored["ops"]) == 26 and stored["configs"][0]["name"] == "c3"
