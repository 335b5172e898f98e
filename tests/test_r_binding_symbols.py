"""The R side of the boundary ships as files under r/ (VERDICT r04 next #6): the drop-in
r/src/vbnmf_update.cpp (same symbol `_ccfindR_vbnmf_update`, arity 4, list keys in the reference's order,
src/RcppExports.cpp:11-32 and src/vbnmf_update.cpp:92-100) and the resident exports r/src/vbnmf_engine.cpp.  R is not in the
image, so they cannot be compiled here; this test keeps them in step with include/vbnmf.h: every vbnmf_* function they call is
declared there with the same number of arguments."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def _calls(text, names):
    """(name, number of arguments) of every call of a declared function in `text` (comments stripped)."""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    found = []
    for mt in re.finditer(r"\b(vbnmf_[a-z0-9_]+)\s*\(", text):
        name = mt.group(1)
        if name not in names:
            continue
        i, depth = mt.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        found.append((name, len(_split_args(text[mt.end():i - 1]))))
    return found


def _declared():
    hdr = open(os.path.join(ROOT, "include", "vbnmf.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    decl = {}
    for mt in re.finditer(r"\b(vbnmf_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = mt.group(2).strip()
        decl[mt.group(1)] = 0 if args in ("", "void") else len(_split_args(args))
    return decl


def test_every_c_abi_call_of_the_r_sources_is_declared_with_the_same_arity():
    decl = _declared()
    assert len(decl) > 60
    seen = set()
    for fn in ("vbnmf_update.cpp", "vbnmf_engine.cpp"):
        text = open(os.path.join(ROOT, "r", "src", fn)).read()
        # exported R-level functions of the same prefix defined IN these files are not C-ABI calls
        local = set(re.findall(r"^(?:SEXP|void|int|double|Rcpp::\w+)\s+(vbnmf_[a-z0-9_]+)\s*\(", text, flags=re.M))
        for name, nargs in _calls(text, set(decl) - local):
            assert decl[name] == nargs, (fn, name, nargs, decl[name])
            seen.add(name)
    for must in ("vbnmf_update_dense", "vbnmf_last_error", "vbnmf_engine_create_geom", "vbnmf_engine_set_state", "vbnmf_engine_step",
                 "vbnmf_engine_run", "vbnmf_engine_get_state", "vbnmf_matrix_from_csc", "vbnmf_matrix_from_dense", "vbnmf_matrix_from_mtx",
                 "vbnmf_engine_ml_step", "vbnmf_comm_create", "vbnmf_engine_attach_comm", "vbnmf_engine_allreduce", "vbnmf_batch_run",
                 "vbnmf_set_engine_grid", "vbnmf_batch_ml_run", "vbnmf_set_engine_padding"):
        assert must in seen, must


def test_the_drop_in_keeps_the_reference_symbol_arity_and_list_keys():
    text = open(os.path.join(ROOT, "r", "src", "vbnmf_update.cpp")).read()
    assert re.search(r'\{"_ccfindR_vbnmf_update",\s*\(DL_FUNC\)\s*&_ccfindR_vbnmf_update,\s*4\}', text)
    assert "R_useDynamicSymbols(dll, FALSE)" in text
    keys = re.findall(r'Rcpp::Named\("(\w+)"\)', text[text.index("Rcpp::List::create"):])
    assert keys[:9] == ["w", "h", "lw", "lh", "ew", "eh", "lkh", "dw", "dh"]          # src/vbnmf_update.cpp:92-100, in that order
    mk = open(os.path.join(ROOT, "r", "src", "Makevars")).read()
    libs = next(ln for ln in mk.splitlines() if ln.startswith("PKG_LIBS"))
    assert "-lvbnmf_hip" in libs and "gsl" not in libs                              # (the reference's src/Makevars:2 links GSL)
