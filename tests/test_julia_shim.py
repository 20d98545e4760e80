"""Guards julia/CarParkingMapsAMD.jl, which no Julia has ever parsed (there is no Julia runtime in the build image nor on the GPU
box): every `ccall((:cpm_..., libcpm), Ret, (types...), args...)` in it is checked mechanically against the prototype of the same
symbol in include/cpm.h -- symbol exists, arity, every argument's C type, the return type -- and the four overriding definitions
keep the positional signatures main.jl calls them with (main.jl:82,85,91,95 of the reference, transcribed below; read from the
reference itself too when it is present)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "julia", "CarParkingMapsAMD.jl")
HEADER = os.path.join(ROOT, "include", "cpm.h")

# Julia ccall type -> the C parameter types it may stand for (const dropped, spaces normalised)
JULIA_TO_C = {
    "Cint": {"int32_t"},
    "Int64": {"int64_t"},
    "UInt64": {"uint64_t"},
    "UInt32": {"uint32_t"},
    "Float64": {"double"},
    "Cstring": {"char*"},
    "Ptr{Cvoid}": {"cpm_ctx*", "void*"},
    "Ptr{Float64}": {"double*"},
    "Ptr{Int64}": {"int64_t*"},
    "Ptr{UInt64}": {"uint64_t*"},
    "Ref{Int64}": {"int64_t*"},
    "Ref{Cint}": {"int32_t*"},
    "Ref{Ptr{Cvoid}}": {"cpm_ctx**", "void**"},
}
JULIA_RET = {"Cint": "int32_t", "Cstring": "char*"}


def _split_top(s):
    """split at commas that are not inside (), {} or []"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(text, start):
    """text[start] == '(' -> index just past its matching ')'"""
    depth = 0
    for i in range(start, len(text)):
        if text[i] == "(":
            depth += 1
        elif text[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced parentheses")


def julia_ccalls(src):
    calls = []
    for m in re.finditer(r"ccall\(", src):
        end = _balanced(src, m.end() - 1)
        parts = _split_top(src[m.end():end - 1])
        sym = re.fullmatch(r"\(:(\w+),\s*libcpm\)", parts[0])
        assert sym, parts[0]
        argt = parts[2].strip()
        assert argt.startswith("(") and argt.endswith(")"), argt
        types = _split_top(argt[1:-1])
        line = src.count("\n", 0, m.start()) + 1
        calls.append(dict(symbol=sym.group(1), ret=parts[1].strip(), types=types, nargs=len(parts) - 3, line=line))
    return calls


def c_prototypes(header):
    text = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int32_t)\s*(cpm_\w+)\s*\(([^)]*)\)\s*;", text):
        ret = "char*" if "char" in m.group(1) else "int32_t"
        params = []
        raw = m.group(3).strip()
        if raw and raw != "void":
            for prm in raw.split(","):
                t = re.sub(r"\bconst\b", "", prm).strip()
                t = re.sub(r"\s*\w+$", "", t) if not t.endswith("*") else t      # drop the parameter name
                stars = t.count("*")
                base = t.replace("*", "").strip()
                params.append(base + "*" * stars)
        protos[m.group(2)] = (ret, params)
    return protos


def test_every_ccall_matches_its_prototype():
    src = open(SHIM).read()
    protos = c_prototypes(open(HEADER).read())
    calls = julia_ccalls(src)
    assert len(calls) >= 15
    assert len(protos) >= 35 and "cpm_resample" in protos and protos["cpm_create"][1] == ["cpm_ctx**", "int64_t", "int64_t", "int32_t"]
    for c in calls:
        where = f"julia/CarParkingMapsAMD.jl:{c['line']} {c['symbol']}"
        assert c["symbol"] in protos, f"{where}: not declared in include/cpm.h"
        ret, params = protos[c["symbol"]]
        assert JULIA_RET.get(c["ret"]) == ret, f"{where}: returns {c['ret']}, header says {ret}"
        assert len(c["types"]) == len(params), f"{where}: {len(c['types'])} argument types, header has {len(params)}"
        assert c["nargs"] == len(params), f"{where}: {c['nargs']} arguments passed for {len(params)} parameters"
        for k, (jt, ct) in enumerate(zip(c["types"], params)):
            assert jt in JULIA_TO_C, f"{where}: argument {k + 1}: no rule for Julia type {jt}"
            assert ct in JULIA_TO_C[jt], f"{where}: argument {k + 1}: {jt} against {ct}"


# the calls main.jl makes (reference main.jl:82, 85, 91, 95): function -> number of positional arguments
MAIN_JL_CALLS = {
    "createpdrive": ["datamatrix", "distance_matrix_km", "number_zones"],
    "createpdestin": ["datamatrix", "number_zones"],
    "solveinitialvalueproblem": ["state_matrix", "transition_matrix", "p_drive", "p_dest", "C", "number_zones"],
    "resampling": ["state_matrix", "transition_matrix", "C", "number_zones", "p_drive", "p_dest", "datamatrix", "distance_matrix_km"],
}


def test_the_four_overrides_keep_main_jls_positional_signatures():
    src = open(SHIM).read()
    tail = src[src.index("end # module"):]
    for name, args in MAIN_JL_CALLS.items():
        m = re.search(rf"^{name}\(([^)]*)\)\s*=\s*CarParkingMapsAMD\.{name}\(([^)]*)\)", tail, flags=re.M)
        assert m, f"override of {name} missing behind the module"
        outer, inner = _split_top(m.group(1)), _split_top(m.group(2))
        assert len(outer) == len(args) and outer == inner, (name, outer, inner)
        inside = re.search(rf"^function {name}\(([^)]*)\)", src, flags=re.M)
        assert inside, name
        names = [re.sub(r"::.*", "", a).strip() for a in _split_top(inside.group(1))]
        assert names == args, (name, names)
    ref = "/root/reference/main.jl"          # present in the build container only
    if os.path.exists(ref):
        text = open(ref).read()
        for name, args in MAIN_JL_CALLS.items():
            m = re.search(rf"=\s*{name}\(([^)]*)\)", text)
            assert m and [a.strip() for a in m.group(1).split(",")] == args, name


def test_the_parser_catches_a_drifted_signature():
    protos = c_prototypes(open(HEADER).read())
    bad = "_check(ccall((:cpm_set_p_drive, libcpm), Cint, (Ptr{Cvoid}, Ptr{Int64}), c.h, p_drive))"
    c = julia_ccalls(bad)[0]
    assert protos[c["symbol"]][1] == ["cpm_ctx*", "double*"]
    assert protos[c["symbol"]][1][1] not in JULIA_TO_C[c["types"][1]]
    short = julia_ccalls("ccall((:cpm_init_states, libcpm), Cint, (Ptr{Cvoid}, Int64, Int64, Int64), c.h, C, cpz, 0)")[0]
    assert len(short["types"]) != len(protos["cpm_init_states"][1])
