"""Compiler-output guards for the dominant kernel (speechclip_amd/csrc/gemm8p.hip), checked on the gfx950 ISA hipcc emits (no GPU needed):
 * no persistent-kernel variant uses scratch (a spill inside the k-loop costs more than any schedule gains: round-5 log, EXPERIMENTS.md R5-3);
 * the register that receives the tile counter's `global_atomic_add` result is in flight for a whole k-step behind an inline-asm issue the compiler
   knows nothing about: between its initialisation, the atomic and the `ds_write_b32` that hands it to the other waves nothing may read or write it;
 * the residual epilogues contain exactly the counted waits that were designed (no compiler-inserted vmcnt(0) between residual uses)."""
import os
import re
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "gemm8p.s"
    src = os.path.join(ROOT, "speechclip_amd", "csrc", "gemm8p.hip")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-S", "--cuda-device-only", src, "-o", str(out)],
                   check=True, capture_output=True, timeout=600)
    txt = out.read_text()
    ks = {}
    for part in re.split(r"\n(?=_ZN12_GLOBAL__N_118gemm8p_pers_kernel\S+:)", txt)[1:]:
        name = part.split(":", 1)[0]
        ks[name] = part.split(".end_amdhsa_kernel")[0]
    assert len(ks) == 24, sorted(ks)                       # ACT x RES x F32 x F16
    return ks


def test_no_variant_spills(kernels):
    for name, body in kernels.items():
        m = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        assert m and int(m.group(1)) == 0, (name, m and m.group(1))
        assert "scratch_" not in body, name
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)) <= 256


def test_tile_counter_result_register_is_untouched_while_in_flight(kernels):
    for name, body in kernels.items():
        # (round 6: 64-bit counter words, generation in the high half: global_atomic_add_x2 into a register PAIR; the low register goes to LDS)
        m = re.search(r"global_atomic_add_x2 v\[(\d+):(\d+)\], v\[\d+:\d+\], v\[\d+:\d+\], off sc0", body)
        assert m, name
        lo, hi = int(m.group(1)), int(m.group(2))
        assert hi == lo + 1

        def touches(line):
            if re.search(r"\bv%d\b|\bv%d\b" % (lo, hi), line):
                return True
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", line):
                if int(a) <= hi and int(b) >= lo:
                    return True
            return False
        lines = [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(";")]
        at = [k for k, l in enumerate(lines) if l.startswith("global_atomic_add_x2")]
        assert len(at) == 1, name
        at = at[0]
        # forward: nothing touches the pair until the ds_write_b32 that hands the low word to the other waves
        fwd = [l for l in lines[at + 1:] if touches(l)]
        assert fwd and fwd[0].startswith("ds_write_b32"), (name, fwd[:3])
        # backward: the nearest instructions touching the pair are its initialisation (one v_mov_b64 or two v_mov_b32), nothing between them and the atomic
        back = [l for l in reversed(lines[:at]) if touches(l)][:2]
        kinds = [l.split()[0] for l in back]
        assert kinds and kinds[0] in ("v_mov_b32_e32", "v_mov_b64_e32"), (name, back)
        if kinds[0] == "v_mov_b32_e32":
            assert len(kinds) == 2 and kinds[1] == "v_mov_b32_e32", (name, back)


def test_residual_epilogues_wait_by_count_only(kernels):
    """Behind the last MFMA of a residual variant: the explicit residual loads, then waits vmcnt(6 6 6 6 6 4 2 0) (bf16: 4 row blocks x 2 loads in flight) or
    vmcnt(4 x 7, 0) (fp32: 2 row blocks x 4 loads), then the stores -- and no other vmcnt wait in between."""
    for name, body in kernels.items():
        m = re.search(r"ILi(\d)ELb([01])ELb([01])ELb([01])EEEv12Gemm8pParams$", name)      # <ACT, RES, F32, F16>
        assert m, name
        if m.group(2) != "1":                               # RES == false
            continue
        f32 = m.group(3) == "1"
        tail = body[body.rfind("v_mfma"):]
        first_store = tail.find("global_store_dwordx4")
        waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", tail[:first_store])]
        want = [4] * 7 + [0] if f32 else [6] * 5 + [4, 2, 0]
        assert waits[-len(want):] == want, (name, waits)
        assert len(re.findall(r"global_load_dwordx4", tail[:first_store])) == (32 if f32 else 16), name
