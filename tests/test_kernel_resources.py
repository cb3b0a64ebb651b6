"""CPU: compile the resident kernel to gfx950 ISA (no GPU needed) and check the properties the
design depends on: the tree stays in VGPRs (no spill, no demotion of the register array to a
dynamically indexed scratch array), the arithmetic is unfused, the launch geometry fits the CU."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "oxmpl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def resident_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asm") / "rrt_resident.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "rrt_resident.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm):
    """name -> metadata dict from the amdhsa.kernels YAML block"""
    meta = {}
    for block in asm.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        meta[name] = {k: int(v) for k, v in re.findall(r"\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|"
                                                        r"private_segment_fixed_size|group_segment_fixed_size|"
                                                        r"max_flat_workgroup_size):\s+(\d+)", block)}
    return meta


def test_resident_tree_stays_in_registers(resident_asm):
    meta = {k: v for k, v in _kernels(resident_asm).items() if "rrt_resident_kernel" in k}
    assert len(meta) == 8  # dim {2,3} x slots {4,21} x {product, stamped diagnostic}
    for name, m in meta.items():
        assert m["vgpr_spill_count"] == 0, name
        # a demoted tr[DIM][S] array would need >= 8*DIM*S bytes of scratch (>= 192 B); the fixed 40 B is
        # the ChaCha block's word buffer
        assert m["private_segment_fixed_size"] <= 64, (name, m)
        assert m["max_flat_workgroup_size"] == 576
        # 9 waves per CU -> at most 3 on a SIMD -> 512/3 = 170 registers per lane
        assert m["vgpr_count"] <= 168, (name, m)
        # 14.4 KB of pipeline state + 576 x 24 B the backend promotes from a small private array
        assert m["group_segment_fixed_size"] <= 32 * 1024
    big = [m for k, m in meta.items() if "Li3ELi21ELb0" in k][0]
    assert big["vgpr_count"] >= 126  # 21 slots x 3 x f64 = 126 VGPRs of tree alone


def test_resident_scan_arithmetic_is_unfused(resident_asm):
    body = resident_asm.split("rrt_resident_kernelILi3ELi21ELb0EEEvNS_9DevParamsE:")[1].split("s_endpgm")[0]
    n_mul, n_add, n_fma = body.count("v_mul_f64"), body.count("v_add_f64"), body.count("v_fma_f64")
    assert n_mul > 200 and n_add > 300          # the unrolled sub/mul/add scan
    assert n_fma < n_mul // 4                   # FMAs appear only inside the sqrt / division expansions
    assert "v_med3_u32" in body and "row_bcast:31" in body and "s_setprio" in body


@pytest.fixture(scope="module")
def lanes_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asm") / "rrt_lanes.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "rrt_lanes.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def test_lanes_headline_kernel_keeps_its_shape(lanes_asm):
    """the kernel KERNEL_AUTO runs for BASELINE.json configs[1], rrt_lanes_kernel<3,24,16,false>: the tree stays in VGPRs (no
    spill, no scratch), three waves fit a SIMD, and the screen is what DESIGN.md 5.5 prices: per PAIR of 64-node register
    rows and query D = 3 packed fused multiply-adds (the query's coordinate broadcast by op_sel on the second operand:
    12 row pairs x 8 queries x 3 = 288) and ONE v_min3_f32 (96) -- 2.0 instructions per (row, query) -- with nothing
    between them that moves data across lanes or touches memory"""
    meta = {k: v for k, v in _kernels(lanes_asm).items() if "rrt_lanes_kernel" in k}
    assert len(meta) >= 12   # R^2..R^6 x row counts x {product, stamped diagnostic}
    name = [k for k in meta if "ILi3ELi24ELi16ELb0" in k][0]
    m = meta[name]
    assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] <= 64, m   # (a demoted tree would need kilobytes of scratch)
    assert m["max_flat_workgroup_size"] == 640 and m["vgpr_count"] <= 168     # 10 waves per CU -> 3 on two of the SIMDs -> 512 / 3
    assert m["group_segment_fixed_size"] <= 64 * 1024
    for k, mm in meta.items():   # R^2 / R^3 at 10,240 nodes: no spill anywhere in the product kernels
        if "Lb0" in k and any(t in k for t in ("ILi2ELi24ELi16", "ILi3ELi24ELi16")):
            assert mm["vgpr_spill_count"] == 0, k
    body = lanes_asm.split(name + ":")[1].split("s_endpgm")[0]
    lines = [l for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    # the scanners' v_min3_f32 are the hand-placed ones; the screen is the runs of code around them (the compiler lays the
    # rows every wave holds and the rows only the six heavy waves hold out as separate blocks)
    mins = [i for i, l in enumerate(lines) if "v_min3_f32" in l]
    clusters, cur = [], [mins[0]]
    for i in mins[1:]:
        if i - cur[-1] > 30:
            clusters.append(cur)
            cur = [i]
        else:
            cur.append(i)
    clusters.append(cur)
    clusters = [c for c in clusters if len(c) >= 16]
    assert sum(len(c) for c in clusters) == 96, [len(c) for c in clusters]
    n_fma = 0
    for c in clusters:
        first = c[0]
        while first > 0 and c[0] - first < 16 and "v_pk_fma_f32" in "".join(lines[first - 3:first]):   # the fused multiply-adds leading up to the first minimum
            first -= 1
        screen = lines[first:c[-1] + 1]
        text = "\n".join(screen)
        n_fma += text.count("v_pk_fma_f32")
        assert "op_sel:[0,1,0]" in text and "op_sel_hi:[1,0,1]" in text       # the query pair's upper / lower half, broadcast
        for bad in ("v_pk_add_f32", "v_pk_mul_f32", "v_and_or_b32", "v_med3_u32", "scratch_", "global_load", "ds_bpermute", "v_readlane",
                    "v_min_f32", "v_writelane"):
            assert bad not in text, bad
        others = [l for l in screen if not any(t in l for t in ("v_pk_fma_f32", "v_min3_f32", "s_waitcnt", "s_nop", "s_cmp", "s_cbranch",
                                                                   "s_min_u32", ".LBB", "v_mov_b32"))]
        assert len(others) <= 8, others
    assert n_fma == 288, n_fma
    assert "v_min_f32_dpp" in body and "s_setprio" in body
    # the resolver's exact work is unfused binary64 (the reference never fuses): sub / mul / add, no v_fma_f64 outside sqrt / division
    assert body.count("v_mul_f64") > 100 and body.count("v_add_f64") > 100


@pytest.fixture(scope="module")
def prm_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asm") / "prm_kernels.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "prm_kernels.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def test_prm_pair_search_streams_the_i_side_through_scalar_loads(prm_asm):
    """the design of DESIGN.md section 9: milestones i are wave-uniform scalar operands (s_load of the fl32 shadow), the
    screen is packed binary32 arithmetic, the loop touches neither LDS nor vector memory, nothing spills, and the exact
    binary64 recheck of the screen's hits is unfused"""
    meta = {k: v for k, v in _kernels(prm_asm).items() if "prm_pairs_kernel" in k}
    assert len(meta) == 8                      # dim 1..8
    for name, m in meta.items():
        assert m["vgpr_spill_count"] == 0 and m["private_segment_fixed_size"] == 0, name
        assert m["max_flat_workgroup_size"] == 256
    body = re.split(r"prm_pairs_kernelILi6EEEvNS_7PrmArgsE\w*: ; @", prm_asm)[1].split("s_endpgm")[0]
    assert body.count("s_load_dwordx4") >= 2 and body.count("s_load_dwordx2") >= 2   # 6 floats per i, two register sets in flight
    assert "v_fma_f64" not in body
    # 4 milestones per thread = 2 packed registers per coordinate; per i and loop body: 12 subtractions, 2 squares, 10 fmas
    assert body.count("v_pk_add_f32") >= 40 and body.count("v_pk_fma_f32") >= 36
    # the recheck: 6 dims of sub / mul / add per hit, unfused (rvss.rs:137-155)
    assert body.count("v_mul_f64") >= 24 and body.count("v_add_f64") >= 44
    # the i side never goes through LDS: the only LDS traffic is the staging buffer of the (rare) hits
    loop = body.split("sched_barrier")[1]
    assert "ds_read" not in loop.split("s_cbranch")[0]


@pytest.fixture(scope="module")
def star_wire_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asm") / "rrt_star_wire.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "rrt_star_wire.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def test_rrt_star_pair_search_keeps_its_shape(star_wire_asm):
    """the neighbour search of the decoupled RRT* (DESIGN.md 10.1), star_pairs_kernel<3, count>: the wave-uniform fl32 rows
    come through SCALAR loads although the kernel also stores
    chunks and bumps a cursor -- as ordinary global loads the compiler turns them into per-lane vector loads and the pass
    takes twice as long (measured: 26.7 ms against 14.4) --, the screen is packed binary32 (16 node pairs x 6 instructions per
    trip), nothing spills, and eight waves fit a SIMD; the wiring kernel keeps its prefetched entries in registers"""
    meta = _kernels(star_wire_asm)
    names = [k for k in meta if "star_pairs_kernelILi3ELb0E" in k]
    assert len(names) == 1
    m = meta[names[0]]
    assert m["vgpr_count"] <= 64 and m["vgpr_spill_count"] <= 1 and m["group_segment_fixed_size"] <= 16 * 1024, m
    body = star_wire_asm.split(names[0] + ":")[1].split("s_endpgm")[0]
    # 32 nodes x 3 coordinates x 4 bytes per trip arrive through scalar loads (however the compiler groups them) ...
    scalar_dwords = sum(n * body.count("s_load_dword" + w + " ") for w, n in (("", 1), ("x2", 2), ("x4", 4), ("x8", 8), ("x16", 16)))
    assert scalar_dwords >= 96, scalar_dwords
    assert "global_load_dwordx4" not in body          # ... and not in the per-lane form the rows must not take
    packed = body.count("v_pk_add_f32") + body.count("v_pk_mul_f32") + body.count("v_pk_fma_f32")
    assert packed >= 96, packed
    assert body.count("v_mul_f64") >= 3 and body.count("v_add_f64") >= 5   # the exact test: unfused binary64 sub / mul / add (fmas only inside sqrt)
    wire = [k for k in meta if "star_wire_kernel" in k]
    assert len(wire) == 1 and meta[wire[0]]["private_segment_fixed_size"] == 0
    wbody = star_wire_asm.split(wire[0] + ":")[1].split("s_endpgm")[0]
    for bad in ("scratch_", "flat_load", "ds_write", "ds_read"):   # entries in registers: no struct bounced through memory
        assert bad not in wbody, bad


@pytest.fixture(scope="module")
def cells_asm(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asm") / "rrt_cells.s")
    subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                           "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "rrt_cells.hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


def test_cells_headline_kernel_keeps_its_shape(cells_asm):
    """rrt_cells_kernel<3, false, FROZEN>: the frozen specialisation fits three waves per SIMD (<= 168 registers,
    at most a handful of spilled values outside the rounds) and three workgroups' LDS per CU, the growing one two waves without
    spilling; a block's entries go through the packed binary32 pipe; the sampler's window, the staged nodes and the tail list
    are LDS; neither hot loop flushes a cache (the device-scope release fences left are the ones around a rebuild, before the
    literal tie loop and at the end of a growing launch)"""
    meta = _kernels(cells_asm)
    frozen = [k for k in meta if "rrt_cells_kernelILi3ELb0ELb1E" in k]
    grow = [k for k in meta if "rrt_cells_kernelILi3ELb0ELb0E" in k]
    assert len(frozen) == 1 and len(grow) == 1
    f, g = meta[frozen[0]], meta[grow[0]]
    assert f["vgpr_count"] <= 168 and f["vgpr_spill_count"] <= 24 and f["max_flat_workgroup_size"] == 256, f
    assert g["vgpr_count"] <= 256 and g["vgpr_spill_count"] == 0 and g["private_segment_fixed_size"] == 0, g
    assert f["group_segment_fixed_size"] <= 160 * 1024 // 3 and g["group_segment_fixed_size"] <= 160 * 1024 // 2
    fbody = cells_asm.split(frozen[0] + ":")[1].split("s_endpgm")[0]
    gbody = cells_asm.split(grow[0] + ":")[1].split("s_endpgm")[0]
    for body in (fbody, gbody):
        assert body.count("v_pk_fma_f32") >= 100 and body.count("v_pk_mul_f32") >= 20
        assert body.count("v_cvt_f32_u32_sdwa") >= 100          # 16-bit coordinates converted straight out of their words
        assert "ds_bpermute_b32" in body                        # the tail pass fetches its pair's query across lanes
    # L2 write-backs (buffer_wbl2): none in a frozen launch's code but the tie loop's, few in the growing kernel
    assert fbody.count("buffer_wbl2") <= 1, fbody.count("buffer_wbl2")
    assert gbody.count("buffer_wbl2") <= 6, gbody.count("buffer_wbl2")
    # the resolver's exact work is unfused binary64
    assert fbody.count("v_mul_f64") > 50 and fbody.count("v_add_f64") > 50


@pytest.fixture(scope="module")
def connect_asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("asm_connect")
    out = {}
    for src in ("rrt_connect.hip", "rrt_connect_se2.hip"):
        path = str(d / (src[:-4] + ".s"))
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                               "-S", "--cuda-device-only", "-o", path, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
        out[src] = open(path).read()
    return out


def test_rrtconnect_kernels_keep_one_wave_per_simd(connect_asm):
    """rrt_connect.hip / rrt_connect_se2.hip (round 3): one wave per problem, four problems per CU -- 40 KB of LDS at most (the SE(2)
    shape for large batches: seven per CU), no scratch (a q_new[lane] once became a scratch array), and the tree / segment / sphere
    tables are read through LDS or global instructions, never flat ones (a select of an LDS and a global address)"""
    se2 = {k: v for k, v in _kernels(connect_asm["rrt_connect_se2.hip"]).items() if "rrt_connect_se2_kernel" in k}
    assert len(se2) == 5   # (768, LDS table) x {product, stamped}, (768, HBM table) x {product, stamped}, (512, HBM table)
    for name, m in se2.items():
        assert m["max_flat_workgroup_size"] == 64 and m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (name, m)
        assert m["group_segment_fixed_size"] <= (23405 if "ILi512E" in name else 40960), (name, m)
    rn = {k: v for k, v in _kernels(connect_asm["rrt_connect.hip"]).items() if "rrt_connect_kernel" in k}
    assert len(rn) == 14   # dim {2 .. 7, runtime} x obstacle table {LDS, HBM}
    for name, m in rn.items():
        assert m["max_flat_workgroup_size"] == 64 and m["private_segment_fixed_size"] == 0 and m["vgpr_spill_count"] == 0, (name, m)
        assert m["group_segment_fixed_size"] <= 40960, (name, m)
    body = connect_asm["rrt_connect_se2.hip"].split("rrt_connect_se2_kernelILi768ELb1ELb0EEEvNS_9DevParamsE:")[1].split("s_endpgm")[0]
    assert "flat_load" not in body and "ds_read_b128" in body and "v_med3_f32" in body   # the shadow scan: LDS reads, top-2 by min / med3
    assert body.count("s_mul_hi_u32") >= 8            # the checksum's folds on the scalar unit
    body3 = connect_asm["rrt_connect.hip"].split("rrt_connect_kernelILi3ELb1EEEvNS_9DevParamsE:")[1].split("s_endpgm")[0]
    assert "flat_load" not in body3
