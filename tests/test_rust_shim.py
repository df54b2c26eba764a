"""The Rust binding under shim/rust/ cannot be compiled in this image (no rustc / cargo), so it is held to the C header
mechanically: every export of include/friedrich_amd.h appears in shim/rust/src/algebra/ffi.rs, in the header's order, with the
header's argument list (C types mapped to their Rust FFI spelling); every `fr_*` call of the other shim files has the declared
number of arguments; the constants agree; the panic texts are the reference's (src/algebra/mod.rs:85,90;
src/gaussian_process/mod.rs:150,203,263,345; multivariate_normal.rs:57; prior.rs:149)."""
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "friedrich_amd.h")
SHIM = os.path.join(ROOT, "shim", "rust")
FFI = os.path.join(SHIM, "src", "algebra", "ffi.rs")

C2RUST = {
    "int": "c_int", "int64_t": "i64", "double": "f64",
    "fr_ctx*": "*mut fr_ctx", "const fr_ctx*": "*const fr_ctx", "fr_ctx**": "*mut *mut fr_ctx",
    "fr_chol*": "*mut fr_chol", "const fr_chol*": "*const fr_chol", "fr_chol**": "*mut *mut fr_chol",
    "const fr_kprog*": "*const fr_kprog",
    "double*": "*mut f64", "const double*": "*const f64", "double**": "*mut *mut f64",
    "int64_t*": "*mut i64", "int*": "*mut c_int",
    "const char*": "*const c_char", "void*": "*mut c_void", "const void*": "*const c_void",
}
RET = {"int": "c_int", "void": None, "const char*": "*const c_char"}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _c_type(arg):
    """'const double* Xq' -> 'const double*'"""
    arg = " ".join(arg.split())
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", arg)
    assert m, arg
    t = m.group(1).strip()
    t = re.sub(r"\s*\*", "*", t)
    return t


def header_exports():
    text = _strip_comments(open(HEADER).read())
    out = []
    for m in re.finditer(r"(?m)^\s*(int|void|const char\s*\*)\s+(fr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text):
        ret = re.sub(r"\s*\*", "*", m.group(1))
        args = [a.strip() for a in m.group(3).split(",")] if m.group(3).strip() not in ("", "void") else []
        out.append((m.group(2), RET[ret], [C2RUST[_c_type(a)] for a in args]))
    return out


def rust_exports():
    text = _strip_comments(open(FFI).read())
    block = text[text.index('extern "C"'):]
    out = []
    for m in re.finditer(r"pub fn (fr_[a-z0-9_]+)\s*\(([^)]*)\)\s*(->\s*([^;]+))?;", block):
        args = []
        for a in [x.strip() for x in m.group(2).split(",") if x.strip()]:
            name, typ = a.split(":", 1)
            args.append(" ".join(typ.split()))
        ret = " ".join(m.group(4).split()) if m.group(4) else None
        out.append((m.group(1), ret, args))
    return out


def test_every_export_is_bound_in_order_with_the_headers_argument_list():
    h, r = header_exports(), rust_exports()
    assert len(h) >= 44, len(h)
    assert [x[0] for x in r] == [x[0] for x in h]
    for (name, hret, hargs), (_, rret, rargs) in zip(h, r):
        assert rret == hret, (name, hret, rret)
        assert rargs == hargs, (name, hargs, rargs)


def test_constants_match_the_header():
    htext = _strip_comments(open(HEADER).read())
    rtext = open(FFI).read()
    consts = dict(re.findall(r"\b(FR_[A-Z0-9_]+)\s*=\s*(-?\d+)", htext))
    consts.update(dict(re.findall(r"#define\s+(FR_[A-Z0-9_]+)\s+(-?\d+)", htext)))
    assert {"FR_OK", "FR_NOT_POSITIVE_DEFINITE", "FR_K_SQUAREDEXP", "FR_K_PROD", "FR_LAYOUT_ROWPTRS", "FR_KPROG_MAX_OPS",
            "FR_ABI_VERSION", "FR_COMM_ID_BYTES"} <= set(consts)
    rconsts = dict(re.findall(r"pub const (FR_[A-Z0-9_]+):\s*[a-z_0-9]+\s*=\s*(-?\d+);", rtext))
    for k, v in consts.items():
        if k in ("FRIEDRICH_AMD_H", "FR_PROF_COUNT"):
            continue
        assert rconsts.get(k) == v, (k, v, rconsts.get(k))
    # struct layouts: fr_kernel_op {i32, i32, [f64; 3]}, fr_kprog {i32, i32, [fr_kernel_op; 15]}
    assert re.search(r"pub struct fr_kernel_op\s*\{\s*pub kind: i32,\s*pub nparams: i32,\s*pub params: \[f64; 3\],\s*\}", rtext)
    assert re.search(r"pub struct fr_kprog\s*\{\s*pub nops: i32,\s*pub reserved: i32,\s*pub ops: \[fr_kernel_op; FR_KPROG_MAX_OPS\],\s*\}", rtext)


def _calls(text, name):
    """argument counts of every call `name(...)` in text (balanced parentheses, top-level commas)"""
    counts = []
    for m in re.finditer(r"\b" + name + r"\s*\(", text):
        i, depth, args, cur = m.end(), 1, 0, ""
        while depth:
            ch = text[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            if depth == 1 and ch == ",":
                args += 1
                cur = ""
            elif depth:
                cur += ch
            i += 1
        counts.append(args + (1 if cur.strip() else 0))
    return counts


def test_every_call_site_passes_the_declared_number_of_arguments():
    decl = {n: len(a) for n, _, a in rust_exports()}
    used = set()
    for dirpath, _, files in os.walk(os.path.join(SHIM, "src")):
        for f in files:
            if not f.endswith(".rs") or f == "ffi.rs":
                continue
            text = _strip_comments(open(os.path.join(dirpath, f)).read())
            for name, nargs in decl.items():
                for c in _calls(text, name):
                    used.add(name)
                    assert c == nargs, (f, name, c, nargs)
    # the entry points friedrich's own call sites need are all exercised by the shim
    need = {"fr_ctx_create", "fr_last_error", "fr_gram", "fr_chol_from_inputs", "fr_chol_refactor", "fr_chol_add_rows", "fr_chol_info",
            "fr_chol_substitutions", "fr_chol_download_l", "fr_chol_upload_l", "fr_chol_free", "fr_chol_set_targets", "fr_likelihood",
            "fr_predict_mean", "fr_predict_variance", "fr_predict_mean_variance", "fr_predict_covariance", "fr_posterior",
            "fr_grad_terms", "fr_mean_pairwise_distance", "fr_linear_prior_fit", "fr_abi_version"}
    assert need <= used, need - used


def test_panic_texts_are_the_references():
    text = ""
    for dirpath, _, files in os.walk(os.path.join(SHIM, "src")):
        for f in files:
            text += open(os.path.join(dirpath, f)).read()
    for s in ("Cholesky decomposition failed even though we used `cholesky_epsilon` value of {cholesky_epsilon}",  # algebra/mod.rs:85
              "Cholesky decomposition failed, consider setting `cholesky_epsilon` via `GaussianProcessBuilder`",   # algebra/mod.rs:90
              "The noise parameter should non-negative but we tried to set it to {}",                              # mod.rs:150
              "likelihood : solve failed",                                                                         # mod.rs:203
              "predict_covariance : solve failed",                                                                 # mod.rs:263, :345
              "MultivariateNormal: Cholesky decomposition failed!",                                                # multivariate_normal.rs:57
              "Linear prior fit : solve failed."):                                                                 # prior.rs:149
        assert s in text, s


def test_all_nine_kernels_and_both_combinators_have_a_device_program():
    text = open(os.path.join(SHIM, "src", "parameters", "kernel_device.rs")).read()
    for kind in ("FR_K_LINEAR", "FR_K_POLYNOMIAL", "FR_K_SQUAREDEXP", "FR_K_EXPONENTIAL", "FR_K_MATERN1", "FR_K_MATERN2", "FR_K_HYPERTAN",
                 "FR_K_MULTIQUADRIC", "FR_K_RATIONALQUADRATIC", "FR_K_SUM", "FR_K_PROD"):
        assert kind in text, kind
    # parameter order = get_parameters() order of the reference (kernel.rs:393, 474, 583, 688, 795, 907, 991, 1061, 1147)
    for frag in ("&[k.c]", "&[k.alpha, k.c, k.d]", "&[k.ls, k.ampl]", "&[k.alpha, k.c]", "&[k.alpha, k.ls]"):
        assert frag in text, frag
