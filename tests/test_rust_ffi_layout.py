"""rust/oxmpl-hip/src/ffi.rs states the byte layout of the two configuration structs of the C ABI.  This test compiles a
C program that prints sizeof / offsetof from include/oxmpl_hip.h and compares: header == ffi.rs tables == the ctypes
structures of oxmpl_amd/capi.py.  It also checks that ffi.rs declares every entry point its extern block needs to exist
in the header (and nothing the header does not have)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FFI = os.path.join(ROOT, "rust", "oxmpl-hip", "src", "ffi.rs")
HEADER = os.path.join(ROOT, "include", "oxmpl_hip.h")


def _rust_table(name):
    src = open(FFI).read()
    body = re.search(r"pub const %s: &\[\(&str, usize, usize\)\] = &\[(.*?)\];" % name, src, re.S).group(1)
    return [(m.group(1), int(m.group(2)), int(m.group(3))) for m in re.finditer(r'\("(\w+)",\s*(\d+),\s*(\d+)\)', body)]


def _rust_const(name):
    return int(re.search(r"pub const %s: usize = (\d+);" % name, open(FFI).read()).group(1))


def _c_layout(tmp_path, struct, fields):
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "oxmpl_hip.h"', 'int main(void) {',
            '  printf("size %%zu\\n", sizeof(%s));' % struct]
    for f in fields:
        prog.append('  printf("%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (f, struct, f, struct, f))
    prog.append('  return 0; }')
    c = tmp_path / ("layout_%s.c" % struct)
    c.write_text("\n".join(prog))
    exe = tmp_path / ("layout_%s" % struct)
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    size = int(out[0].split()[1])
    rows = [(l.split()[0], int(l.split()[1]), int(l.split()[2])) for l in out[1:] if l]
    return size, rows


def test_rrt_config_layout_agrees_everywhere(tmp_path):
    from oxmpl_amd import capi
    table = _rust_table("OXHIP_RRT_CONFIG_LAYOUT")
    size, rows = _c_layout(tmp_path, "oxhip_rrt_config", [f for f, _, _ in table])
    assert rows == table and size == _rust_const("OXHIP_RRT_CONFIG_SIZE")
    import ctypes
    assert ctypes.sizeof(capi.Config) == size
    assert [(n, getattr(capi.Config, n).offset, getattr(capi.Config, n).size) for n, _ in capi.Config._fields_] == table


def test_prm_config_layout_agrees_everywhere(tmp_path):
    from oxmpl_amd import capi
    table = _rust_table("OXHIP_PRM_CONFIG_LAYOUT")
    size, rows = _c_layout(tmp_path, "oxhip_prm_config", [f for f, _, _ in table])
    assert rows == table and size == _rust_const("OXHIP_PRM_CONFIG_SIZE")
    import ctypes
    assert ctypes.sizeof(capi.PrmConfig) == size
    assert [(n, getattr(capi.PrmConfig, n).offset, getattr(capi.PrmConfig, n).size) for n, _ in capi.PrmConfig._fields_] == table


def test_extern_block_names_exist_in_the_header_with_the_same_arity():
    rust = open(FFI).read()
    header = open(HEADER).read()
    header_flat = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    decls = {m.group(1): m.group(2) for m in re.finditer(r"\b(oxhip_\w+)\s*\(([^;{]*?)\)\s*;", header_flat, re.S)}
    externs = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (oxhip_\w+)\s*\((.*?)\)\s*(?:->|;)", rust, re.S)}
    assert len(externs) >= 35
    for name, args in externs.items():
        assert name in decls, name
        n_rust = len([a for a in args.split(",") if a.strip()])
        c_args = decls[name].strip()
        n_c = 0 if c_args in ("", "void") else len([a for a in c_args.split(",") if a.strip()])
        assert n_rust == n_c, (name, n_rust, n_c)
    # status codes and the ABI version are restated in ffi.rs: keep them in step with the header
    for const in ("OXHIP_ERR_TIMEOUT", "OXHIP_ERR_NO_SOLUTION_FOUND", "OXHIP_ERR_PLANNER_UNINITIALISED",
                  "OXHIP_ERR_INVALID_START_STATE", "OXHIP_ERR_UNSAMPLED_STATE_SPACE", "OXHIP_ERR_BAD_ARG",
                  "OXHIP_ERR_UNBOUNDED", "OXHIP_ERR_ZERO_VOLUME", "OXHIP_ERR_CAPACITY", "OXHIP_ERR_HIP", "OXHIP_ERR_NO_DEVICE"):
        c_val = int(re.search(r"\b%s\s*=\s*(\d+)" % const, header).group(1))
        r_val = int(re.search(r"pub const %s: i32 = (\d+);" % const, rust).group(1))
        assert c_val == r_val, const
    assert int(re.search(r"#define OXHIP_ABI_VERSION\s+(\d+)", header).group(1)) == \
        int(re.search(r"pub const OXHIP_ABI_VERSION: i32 = (\d+);", rust).group(1))
