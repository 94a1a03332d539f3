"""The Rust binding under rust/ cannot be compiled here (no rustc), so its agreement with the C header is checked
textually: the generated extern block is up to date, every exported function is declared, every #[repr(C)]
struct lists the header's fields in the header's order with matching widths, and the C side pins the same
layouts with static_asserts."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS_RS = os.path.join(ROOT, "rust", "unet-offload-sys", "src", "lib.rs")
SAFE_RS = os.path.join(ROOT, "rust", "unet-offload", "src", "lib.rs")
HEADER = os.path.join(ROOT, "include", "unet_offload.h")

C2RUST = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64"}
NAMES = {"unet_result": "UnetResult", "unet_iface": "UnetIface", "unet_tx_desc": "UnetTxDesc", "unet_tx_fields": "UnetTxFields",
         "unet_pcap_info": "UnetPcapInfo", "unet_ticket": "UnetTicket", "unet_tcp_flow": "UnetTcpFlow", "unet_tcp_stats": "UnetTcpStats"}


def c_structs():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (unet_[a-z_]+) \{(.*?)\}\s*\1\s*;", text, flags=re.S):
        fields = []
        for f in re.finditer(r"(uint\d+_t)\s+([a-z0-9_]+)(?:\[(\d+)\])?\s*;", m.group(2)):
            fields.append((f.group(2), C2RUST[f.group(1)] if not f.group(3) else f"[{C2RUST[f.group(1)]}; {f.group(3)}]"))
        out[m.group(1)] = fields
    return out


def rust_structs():
    text = open(SYS_RS).read()
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        out[m.group(1)] = [(f.group(1), f.group(2)) for f in re.finditer(r"pub (\w+): ([^,\n]+),", m.group(2))]
    return out


def test_generated_extern_block_is_current():
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]) == 0, \
        "rust/unet-offload-sys/src/lib.rs is stale: run tools/gen_rust_sys.py"


def test_every_export_is_declared_in_rust():
    from test_abi import declared_functions
    rs = open(SYS_RS).read()
    declared = set(re.findall(r"pub fn (unet_[a-z0-9_]+)\(", rs))
    assert declared == set(declared_functions())
    # argument counts agree with the header
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for m in re.finditer(r"\b(unet_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        n_c = 0 if m.group(2).strip() == "void" else m.group(2).count(",") + 1
        r = re.search(r"pub fn " + m.group(1) + r"\(([^)]*)\)", rs)
        n_r = 0 if not r.group(1).strip() else r.group(1).count(",") + 1
        assert n_c == n_r, m.group(1)


def test_repr_c_structs_match_the_header_field_by_field():
    c, r = c_structs(), rust_structs()
    assert set(c) == set(NAMES)
    for cname, rname in NAMES.items():
        assert r[rname] == c[cname], (cname, c[cname], r[rname])


def test_c_side_pins_the_layouts():
    src = open(os.path.join(ROOT, "unet_amd", "csrc", "unet_offload.hip")).read()
    for cname in NAMES:
        assert re.search(r"static_assert\(sizeof\(" + cname + r"\) == \d+", src), cname
    for field in ("unet_result, verdict", "unet_iface, mac", "unet_tx_desc, dst_mac", "unet_tx_fields, l4_csum", "unet_pcap_info, arena_bytes"):
        assert f"offsetof({field})" in src, field


def test_safe_wrapper_uses_only_declared_symbols_and_keeps_the_reference_signature():
    safe = open(SAFE_RS).read()
    rs = open(SYS_RS).read()
    declared = set(re.findall(r"pub (?:fn|const) (\w+)", rs)) | set(re.findall(r"pub struct (\w+)", rs))
    for name in set(re.findall(r"sys::(\w+)", safe)):
        assert name in declared, name
    assert "pub fn calculate_checksum(data: &[u8], sum: u16) -> u16" in safe  # src/utils.rs:1
    codes = dict(re.findall(r"^\s+(\w+) = (\d+),$", safe, flags=re.M))
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    enum = dict((k.replace("UNET_V_", "").replace("_", "").lower(), v) for k, v in re.findall(r"(UNET_V_[A-Z0-9_]+)\s*=\s*(\d+)", header))
    enum.pop("count")
    assert {k.lower(): v for k, v in codes.items()} == enum
    assert safe.count("{") == safe.count("}") and safe.count("(") == safe.count(")")
