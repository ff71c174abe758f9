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
    entry = table["reduce_fused_u8_mfma"]
    assert bench.kernel_isa_sha(entry["symbol"]) == entry["isa_sha"]
    assert bench.traffic_for("reduce_fused_u8_mfma") == entry["traffic_bytes"]
    # the counters it was stamped from are committed, and say what the entry says
    assert os.path.exists(os.path.join(ROOT, entry["profile"]))
    assert entry["traffic_bytes"] == entry["fetch_bytes_x2"] + entry["write_bytes"]
    assert 1.0 < entry["traffic_bytes"] / entry["algorithmic_bytes"] < 1.2


def test_isa_hash_is_of_machine_code_not_of_a_name():
    assert bench.kernel_isa_sha("no_such_kernel_in_the_library") is None
    a = bench.kernel_isa_sha("reduce_fused_u8x4_mfmaILi6ELi1ELi4ELb1ELi0ELb1ELi256ELi1EE")
    b = bench.kernel_isa_sha("reduce_fused_u8x4_mfmaILi6ELi1ELi4ELb0ELi0ELb1ELi256ELi1EE")
    assert a and b and a != b  # (two instantiations of one source: different instructions, different stamps)


def test_ops_table_has_the_calls_of_the_path():
    names = [op["name"] for op in bench.ops_table()]
    assert len(names) == len(set(names))
    for want in ("reducev_8", "reduceh_8", "reduce_rgb_8", "reduce_rgb_7.3", "resize_rgb_to_1000", "thumbnail_500",
                 "shrinkv_4", "shrinkh_4", "convi_3x3_u8", "convi_5x5_u8", "convi_3x3_u16", "gaussblur_s2_u8",
                 "gaussblur_s8_u8", "gaussblur_s2_u16", "gaussblur_s8_u16", "gaussblur_s2_f32",
                 "colourspace_srgb_lab_u8", "colourspace_srgb_labs_u8", "sharpen_u8", "reduce_rgba16_8"):
        assert want in names, want
    for op in bench.ops_table():
        assert ("chain" in op) != ("mask" in op), op["name"]  # every entry has exactly one reference recipe


def test_compact_summary_puts_scalars_where_the_record_keeps_them():
    line = {
        "roofline": {"frac": 0.71, "frac_cold": 0.69, "traffic": 123},
        "configs": [
            {"name": "c3", "ms": 13.0, "frac_fp64": 0.36, "frac_hbm": 0.25, "float_input": {"ms": 14.3},
             "frac_of_fp64_stream": 0.48, "parity": {"bit_exact": True}},
            {"name": "c4", "ms": 43.0, "frac": 0.6, "ms_per_image": 0.042, "images_per_gpu": 1024, "parity": {"bit_exact": True}},
            {"name": "c5", "ms": 140.0, "tflops": 59.0, "frac": 0.75},
        ],
        "ops": [{"name": "reducev_8", "frac": 0.75, "parity": {"bit_exact": True}},
                {"name": "thumbnail_500", "frac": 0.5, "parity": {"bit_exact": False}}],
    }
    bench.compact_summary(line)
    roof = line["roofline"]
    assert roof["c3_ms"] == 13.0 and roof["c3_ms_float_input"] == 14.3 and roof["c4_frac"] == 0.6 and roof["c5_tflops"] == 59.0
    assert roof["op_reducev_8_frac"] == 0.75 and roof["op_thumbnail_500_frac"] == 0.5
    assert all(not isinstance(v, (dict, list)) for k, v in roof.items() if k != "others")
    assert list(line)[-1] == "summary"  # the tail of stdout
    s = line["summary"]
    assert s["c2_frac"] == 0.71 and s["c2_traffic"] == 123 and s["ops_frac"]["reducev_8"] == 0.75
    assert s["parity"] == {"c3": True, "c4": True, "c5": None, "reducev_8": True, "thumbnail_500": False}
    json.dumps(line)
