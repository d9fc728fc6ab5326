#!/usr/bin/env python3
"""Development-time generator for opus_amd/csrc/rfc6716_tables.h.

The Opus bitstream (RFC 6716) is defined by a set of normative constant tables
(PDFs of every entropy-coded symbol, the CELT static allocation / cache tables,
the SILK NLSF codebooks ...).  Any decoder has to hold the same numbers.  This
script reads the numeric literals of those tables out of the reference tree
(src/celt/decoder.rs:90-470, src/silk/tables.rs, src/silk/decoder.rs:445-1293),
re-packs them in this project's own layout (one flat u16 ICDF pool addressed by
offsets, flat row-major integer arrays) and writes a C header.  Only numbers are
taken over; no reference code is parsed for behaviour.

Run once in the build container (needs /root/reference); the generated header is
committed, so nothing at test/bench time reads the reference tree.
"""
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src")
OUT = Path(sys.argv[2]) if len(sys.argv) > 2 else Path(__file__).resolve().parent.parent / "opus_amd" / "csrc" / "rfc6716_tables.h"


class Icdf:
    def __init__(self, total, dist):
        self.total, self.dist = total, list(dist)


TOK = re.compile(r"\s*(::|[A-Za-z_][A-Za-z0-9_]*|0x[0-9A-Fa-f]+|\d+\.\d*(?:[eE][-+]?\d+)?|\d+|.)")


def tokenize(s):
    s = re.sub(r"//[^\n]*", "", s)
    s = re.sub(r"/\*.*?\*/", "", s, flags=re.S)
    return [t for t in TOK.findall(s) if t.strip()]


class Parser:
    def __init__(self, toks, scope):
        self.t, self.i, self.scope = toks, 0, scope

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else None

    def eat(self, x=None):
        v = self.t[self.i]
        if x is not None and v != x:
            raise SyntaxError(f"expected {x} got {v} at {self.i}")
        self.i += 1
        return v

    def value(self):
        while self.peek() == "&":
            self.eat()
        p = self.peek()
        if p == "[":
            self.eat()
            out = []
            while self.peek() != "]":
                out.append(self.value())
                if self.peek() == ",":
                    self.eat()
            self.eat("]")
            return out
        if p == "ICDFContext":
            self.eat()
            self.eat("{")
            self.eat("total"); self.eat(":")
            total = self.value(); self.eat(",")
            self.eat("dist"); self.eat(":")
            dist = self.value()
            if self.peek() == ",":
                self.eat()
            self.eat("}")
            return Icdf(total, dist)
        # scalar expression up to , ] } ;
        expr = []
        depth = 0
        while True:
            p = self.peek()
            if p is None or (depth == 0 and p in (",", "]", "}", ";")):
                break
            if p == "(":
                depth += 1
            if p == ")":
                depth -= 1
            expr.append(self.eat())
        txt = " ".join(expr).replace(" :: ", "__")
        txt = re.sub(r"\bas\s+\w+", "", txt)
        txt = re.sub(r"(\d)(?:f32|i32|u32|usize|i16|u8|i8|u16)\b", r"\1", txt)
        return eval(txt, {}, self.scope)


def parse_consts(path, scope):
    toks = tokenize(path.read_text())
    i, prefix_stack = 0, []
    depth_stack = []
    depth = 0
    while i < len(toks):
        t = toks[i]
        if t == "mod" and i + 2 < len(toks) and toks[i + 2] == "{":
            prefix_stack.append(toks[i + 1]); depth_stack.append(depth)
            depth += 1
            i += 3
            continue
        if t == "{":
            depth += 1
        elif t == "}":
            depth -= 1
            if depth_stack and depth == depth_stack[-1]:
                depth_stack.pop(); prefix_stack.pop()
        if t == "const" and re.match(r"[A-Z]", toks[i + 1]) and toks[i + 2] == ":":
            name = toks[i + 1]
            j = i + 3
            # skip the type up to '=' at bracket depth 0
            d = 0
            while not (toks[j] in ("=", ";") and d == 0):
                if toks[j] in "[(<":
                    d += 1
                if toks[j] in "])>":
                    d -= 1
                j += 1
            if toks[j] == ";" or depth != len(prefix_stack):
                i = j
                continue
            # collect value up to ';' at depth 0
            k, d = j + 1, 0
            while not (toks[k] == ";" and d == 0):
                if toks[k] in "[({":
                    d += 1
                if toks[k] in "])}":
                    d -= 1
                k += 1
            if d == 0 and depth <= len(prefix_stack):  # module-level only
                local = dict(scope)
                pre = "__".join(prefix_stack)
                if pre:
                    for key, v in scope.items():
                        if key.startswith(pre + "__"):
                            local[key[len(pre) + 2:]] = v
                try:
                    val = Parser(toks[j + 1:k], local).value()
                    scope[(pre + "__" if pre else "") + name] = val
                except Exception as e:  # functions' inner consts etc.
                    pass
            i = k
            continue
        i += 1
    return scope


celt = parse_consts(REF / "celt" / "decoder.rs", {})
silk = parse_consts(REF / "silk" / "tables.rs", {})
silk = parse_consts(REF / "silk" / "decoder.rs", silk)

lines = []
pool = []          # flat u16 pool:  total, n, c0..c(n-1)
pool_index = {}


def icdf_off(ic):
    key = (ic.total, tuple(ic.dist))
    if key not in pool_index:
        pool_index[key] = len(pool)
        pool.extend([ic.total, len(ic.dist)] + ic.dist)
    return pool_index[key]


def flat(v):
    if isinstance(v, (list, tuple)):
        out = []
        for x in v:
            out.extend(flat(x))
        return out
    return [v]


def shape(v):
    s = []
    while isinstance(v, (list, tuple)):
        s.append(len(v)); v = v[0]
    return s


def emit_array(ctype, cname, vals, dims=None, fmt="{}", per_line=16):
    dims = dims or [len(vals)]
    d = "".join(f"[{x}]" for x in dims)
    lines.append(f"RFC_TBL {ctype} {cname}{d} = {{")
    row = []
    for i, v in enumerate(vals):
        row.append(fmt.format(v))
        if len(row) == per_line or i == len(vals) - 1:
            lines.append("  " + ", ".join(row) + ",")
            row = []
    lines.append("};")


def emit_int(ctype, cname, v, **kw):
    emit_array(ctype, cname, [int(x) for x in flat(v)], shape(v), **kw)


def emit_icdf_scalar(cname, ic):
    lines.append(f"#define {cname} {icdf_off(ic)}")


def emit_icdf_array(cname, v):
    sh = shape_ic(v)
    emit_array("uint16_t", cname, [icdf_off(x) for x in flat(v)], sh)


def shape_ic(v):
    s = []
    while isinstance(v, (list, tuple)):
        s.append(len(v)); v = v[0]
    return s


def emit_ragged(ctype, cname, v):
    """list of rows with different lengths -> flat data + row offsets"""
    offs, data = [], []
    for row in v:
        offs.append(len(data)); data.extend(int(x) for x in flat(row))
    emit_array(ctype, cname, data)
    emit_array("uint16_t", cname + "_ROW", offs)


H = lines.append
H("/* RFC 6716 normative constant tables, packed for opus_amd (generated by tools/gen_tables.py).")
H(" * ICDF tables live in one pool; an ICDF is addressed by its offset `o`:")
H(" *   RFC_ICDF_POOL[o] = ft (total), [o+1] = n symbols, [o+2+k] = cumulative frequency up to and")
H(" *   including symbol k (so the last entry equals ft).  Multi-dimensional tables are row-major. */")
H("#ifndef OPUS_AMD_RFC6716_TABLES_H")
H("#define OPUS_AMD_RFC6716_TABLES_H")
H("#include <stdint.h>")
H("#ifndef RFC_TBL")
H("#define RFC_TBL static const   /* .hip units redefine this to place tables in device memory */")
H("#endif")
H("")
H("/* ---- CELT (RFC 6716 section 4.3) ---- */")
emit_int("uint8_t", "CELT_EBANDS", celt["FREQ_BANDS"], per_line=22)
emit_int("uint8_t", "CELT_BAND_WIDTH", celt["FREQ_RANGE"], per_line=21)
emit_int("uint8_t", "CELT_LOG_N", celt["LOG_FREQ_RANGE"], per_line=21)
emit_int("uint8_t", "CELT_E_PROB_INTRA", celt["COARSE_ENERGY_INTRA"], per_line=14)
emit_int("uint8_t", "CELT_E_PROB_INTER", celt["COARSE_ENERGY_INTER"], per_line=14)
emit_int("uint8_t", "CELT_CAPS", celt["STATIC_CAPS"], per_line=21)
emit_int("uint8_t", "CELT_ALLOC_VECTORS", celt["STATIC_ALLOC"], per_line=21)
emit_int("uint8_t", "CELT_LOG2_FRAC", celt["LOG2_FRAC"], per_line=24)
emit_int("int8_t", "CELT_TF_SELECT", celt["TF_SELECT"], per_line=8)
emit_int("uint8_t", "CELT_CACHE_BITS", celt["CACHE_BITS"], per_line=20)
emit_int("int16_t", "CELT_CACHE_INDEX", celt["CACHE_INDEX"], per_line=21)
emit_int("uint16_t", "CELT_QN_EXP2", celt["QN_EXP2"])
emit_int("uint8_t", "CELT_BIT_INTERLEAVE", celt["BIT_INTERLEAVE"])
emit_int("uint8_t", "CELT_BIT_DEINTERLEAVE", celt["BIT_DEINTERLEAVE"])
emit_int("uint8_t", "CELT_HADAMARD_ORDERY", celt["HADAMARD_ORDERY"], per_line=30)
emit_int("uint32_t", "CELT_PVQ_U", celt["PVQ_U"], fmt="{}u", per_line=8)
emit_int("uint16_t", "CELT_PVQ_U_ROW", celt["PVQ_U_ROW"], per_line=15)
emit_array("float", "CELT_POSTFILTER_TAPS", [repr(float(x)) + "f" for x in flat(celt["POSTFILTER_TAPS"])], [3, 3])
emit_icdf_scalar("ICDF_CELT_TAPSET", celt["TAPSET"])
emit_icdf_scalar("ICDF_CELT_ENERGY_SMALL", celt["MODEL_ENERGY_SMALL"])
emit_icdf_scalar("ICDF_CELT_SPREAD", celt["MODEL_SPREAD"])
emit_icdf_scalar("ICDF_CELT_ALLOC_TRIM", celt["ALLOC_TRIM"])
H("")
H("/* ---- SILK (RFC 6716 section 4.2) ---- */")
for nm in ["STAGE1", "STAGE2", "STAGE3", "MID_ONLY", "FRAME_TYPE_INACTIVE", "FRAME_TYPE_ACTIVE",
           "LSB_SUBFRAME_GAIN", "DELTA_SUBFRAME_GAIN", "LSF_STAGE2_EXTENSION", "LSF_INTERPOLATION_INDEX",
           "PITCH_DELTA", "PITCH_HIGH_PART", "LTP_PERIODICITY", "LTP_SCALE_INDEX", "LCG_SEED", "EXC_LSB"]:
    pre = "STEREO_" if nm.startswith("STAGE") else ""
    emit_icdf_scalar("ICDF_SILK_" + pre + nm, silk[nm])
emit_icdf_array("ICDF_SILK_MSB_SUBFRAME_GAIN", silk["MSB_SUBFRAME_GAIN"])
emit_icdf_array("ICDF_SILK_LSF_STAGE1_NB_MB", silk["LSF_STAGE1_NB_MB"])
emit_icdf_array("ICDF_SILK_LSF_STAGE1_WB", silk["LSF_STAGE1_WB"])
emit_icdf_array("ICDF_SILK_LSF_STAGE2_NB_MB", silk["lsf_stage2_nb_mb__MAP"])
emit_icdf_array("ICDF_SILK_LSF_STAGE2_WB", silk["lsf_stage2_wb__MAP"])
emit_int("uint8_t", "SILK_LSF_PRED_WEIGHT_NB_MB", silk["LSF_PRED_WEIGHT_NB_MB"], per_line=9)
emit_int("uint8_t", "SILK_LSF_PRED_WEIGHT_WB", silk["LSF_PRED_WEIGHT_WB"], per_line=15)
emit_int("uint8_t", "SILK_LSF_PRED_WEIGHT_INDEX_NB_MB", silk["LSF_PRED_WEIGHT_INDEX_NB_MB"], per_line=9)
emit_int("uint8_t", "SILK_LSF_PRED_WEIGHT_INDEX_WB", silk["LSF_PRED_WEIGHT_INDEX_WB"], per_line=15)
emit_int("uint8_t", "SILK_LSF_CODEBOOK_NB_MB", silk["LSF_CODEBOOK_NB_MB"], per_line=10)
emit_int("uint8_t", "SILK_LSF_CODEBOOK_WB", silk["LSF_CODEBOOK_WB"], per_line=16)
emit_int("uint16_t", "SILK_LSF_WEIGHT_NB_MB", silk["LSF_WEIGHT_NB_MB"], per_line=10)
emit_int("uint16_t", "SILK_LSF_WEIGHT_WB", silk["LSF_WEIGHT_WB"], per_line=16)
emit_int("int16_t", "SILK_LSF_MIN_SPACING_NB_MB", silk["LSF_MIN_SPACING_NB_MB"], per_line=17)
emit_int("int16_t", "SILK_LSF_MIN_SPACING_WB", silk["LSF_MIN_SPACING_WB"], per_line=17)
emit_int("uint8_t", "SILK_LSF_ORDERING_NB_MB", silk["LSF_ORDERING_NB_MB"])
emit_int("uint8_t", "SILK_LSF_ORDERING_WB", silk["LSF_ORDERING_WB"])
emit_int("int16_t", "SILK_COSINE", silk["COSINE"], per_line=12)
emit_icdf_array("ICDF_SILK_PITCH_CONTOUR_NB", silk["PITCH_CONTOUR_NB"])
emit_icdf_array("ICDF_SILK_PITCH_CONTOUR_MB_WB", silk["PITCH_CONTOUR_MB_WB"])
emit_int("int8_t", "SILK_PITCH_OFFSET_NB_10MS", silk["PITCH_OFFSET_NB"][0], per_line=2)
emit_int("int8_t", "SILK_PITCH_OFFSET_NB_20MS", silk["PITCH_OFFSET_NB"][1], per_line=4)
emit_int("int8_t", "SILK_PITCH_OFFSET_MB_WB_10MS", silk["PITCH_OFFSET_MB_WB"][0], per_line=2)
emit_int("int8_t", "SILK_PITCH_OFFSET_MB_WB_20MS", silk["PITCH_OFFSET_MB_WB"][1], per_line=4)
# pitch low-part ICDFs live inside impl blocks (silk/decoder.rs:548-591): uniform over 4/6/8
emit_icdf_array("ICDF_SILK_PITCH_LOW_PART", [Icdf(256, [64, 128, 192, 256]),
                                              Icdf(256, [43, 85, 128, 171, 213, 256]),
                                              Icdf(256, [32, 64, 96, 128, 160, 192, 224, 256])])
emit_icdf_array("ICDF_SILK_LTP_FILTER", silk["LTP_FILTER"])
emit_ragged("int8_t", "SILK_LTP_TAPS", silk["LTP_TAPS"])
emit_int("uint16_t", "SILK_LTP_SCALE", silk["LTP_SCALE"])
emit_icdf_array("ICDF_SILK_EXC_RATE", silk["EXC_RATE"])
emit_icdf_array("ICDF_SILK_PULSE_COUNT", silk["PULSE_COUNT"])
emit_icdf_array("ICDF_SILK_PULSE_LOCATION", silk["PULSE_LOCATION"])
emit_icdf_array("ICDF_SILK_EXC_SIGN", silk["EXC_SIGN"])
emit_int("uint8_t", "SILK_QUANT_OFFSET", silk["QUANT_OFFSET"], per_line=2)
H("")
# pool last (all offsets known now) -- but C needs it before use only for functions, which take the pool by name
emit_array("uint16_t", "RFC_ICDF_POOL", pool, per_line=20)
H("#endif")
OUT.write_text("\n".join(lines) + "\n")
print("wrote", OUT, "pool", len(pool), "entries; celt keys", len(celt), "silk keys", len(silk))
