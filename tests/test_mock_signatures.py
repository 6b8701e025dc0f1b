"""Pins the hand-written MFEM / ExaConstit stand-in (tests/mock_mfem/mock_mfem.hpp) and the adapters (include/exaconstit_mfem_adapters.hpp) to
the reference's own headers: tests/test_adapters.py proves that the adapters work THROUGH BASE-CLASS POINTERS of the mock, which means nothing
if the mock's virtual signatures drift from /root/reference/src/mechanics_model.hpp:70-116,233 (ExaModel) and
/root/reference/src/mechanics_integrators.hpp:14-124 (ExaNLFIntegrator, ICExaNLFIntegrator) - a drifted mock keeps that test green and breaks
the real build.  Here a small C++ declaration reader extracts constructor and virtual / override member signatures (name, return type,
parameter types without names, const-ness, pure-ness) from the reference headers and fails when
  * the mock's ExaModel / ExaNLFIntegrator declare a constructor, virtual or override that the reference class does not have in that exact form,
  * a reference `override` (which the real MFEM base must therefore declare) is missing from the mock's mfem::NonlinearFormIntegrator,
  * an `override` of HipExaModel / HipExaNLFIntegrator (or of the L-vector pair HipExaModelLVec / HipExaNLFIntegratorLVec) has no identical virtual in the
    reference base class,
  * a non-virtual ExaModel member the adapters call (SetModelDt, GetStress1, ...) differs.
Build-container test: the reference tree does not exist on the GPU box (skipped there).  test_checker_notices_a_drifted_mock edits one
signature in memory and requires the checker to go red."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
MOCK = os.path.join(ROOT, "tests", "mock_mfem", "mock_mfem.hpp")
ADAPT = os.path.join(ROOT, "include", "exaconstit_mfem_adapters.hpp")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference headers are only present in the build container")

QUALIFIERS = {"const", "volatile", "struct", "class", "typename"}


def strip_comments(t):
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    return re.sub(r"//[^\n]*", " ", t)


def class_body(text, name):
    m = re.search(r"\bclass\s+" + name + r"\b[^;{]*\{", text)
    assert m, f"class {name} not found"
    i = m.end(); depth = 1
    while depth:
        c = text[i]
        depth += (c == "{") - (c == "}")
        i += 1
    return text[m.end(): i - 1]


def declarations(body):
    """member declarations at depth 0 of a class body; an inline function body ends the declaration it belongs to"""
    out = []; cur = []; i = 0; par = 0
    while i < len(body):
        c = body[i]
        if c == "(":
            par += 1
        elif c == ")":
            par -= 1
        if par == 0 and c == ";":
            out.append("".join(cur)); cur = []
        elif par == 0 and c == "{":
            depth = 1; i += 1
            while depth:
                depth += (body[i] == "{") - (body[i] == "}")
                i += 1
            out.append("".join(cur)); cur = []
            continue
        else:
            cur.append(c)
        i += 1
    return [re.sub(r"\b(public|protected|private)\s*:", " ", d).strip() for d in out if d.strip()]


def norm_type(t):
    t = re.sub(r"\bmfem::", "", t)
    toks = re.findall(r"[A-Za-z_]\w*|::|[*&<>,]", t)
    return " ".join(toks)


def norm_param(p):
    p = p.split("=")[0]
    toks = re.findall(r"[A-Za-z_]\w*|::|[*&<>,]", re.sub(r"\bmfem::", "", p))
    if not toks:
        return ""
    # drop the parameter's name: a trailing identifier, provided a type remains without it
    if re.match(r"[A-Za-z_]", toks[-1]) and any(re.match(r"[A-Za-z_]", x) and x not in QUALIFIERS for x in toks[:-1]):
        toks = toks[:-1]
    return " ".join(toks)


def split_params(s):
    parts = []; cur = []; depth = 0
    for c in s:
        depth += (c in "<(") - (c in ">)")
        if c == "," and depth == 0:
            parts.append("".join(cur)); cur = []
        else:
            cur.append(c)
    if "".join(cur).strip():
        parts.append("".join(cur))
    return [norm_param(p) for p in parts if norm_param(p) not in ("", "void")]


def methods(text, cls):
    """{(name, params): dict(ret, const, pure, virtual, override, ctor)} of the functions class `cls` declares"""
    res = {}
    for d in declarations(class_body(strip_comments(text), cls)):
        if d.startswith("using ") or d.startswith("friend ") or "(" not in d or " operator" in " " + d:
            continue
        m = re.search(r"(~?[A-Za-z_]\w*)\s*\(", d)
        if not m:
            continue
        name = m.group(1)
        i = m.end(); depth = 1
        while depth:
            depth += (d[i] == "(") - (d[i] == ")")
            i += 1
        params = tuple(split_params(d[m.end(): i - 1]))
        head = d[: m.start()]; tail = d[i:].split(":")[0] if name == cls else d[i:]
        head_toks = [x for x in norm_type(head).split() if x not in ("virtual", "inline", "explicit", "static")]
        res[(name, params)] = dict(ret=" ".join(head_toks), const=bool(re.search(r"\bconst\b", tail)), pure=bool(re.search(r"=\s*0", tail)),
                                   virtual="virtual" in head.split(), override=bool(re.search(r"\boverride\b", tail)), ctor=(name == cls))
    return res


def mismatches(mock_text, adapt_text, ref_model, ref_integ):
    """list of human-readable differences; empty = pinned"""
    bad = []
    ref = {"ExaModel": methods(ref_model, "ExaModel"), "ExaNLFIntegrator": methods(ref_integ, "ExaNLFIntegrator"),
           "ICExaNLFIntegrator": methods(ref_integ, "ICExaNLFIntegrator")}
    same = lambda a, b: (a["ret"], a["const"], a["pure"]) == (b["ret"], b["const"], b["pure"])
    # 1. the mock's ExaModel / ExaNLFIntegrator against the reference classes
    for cls, nonvirtual_too in (("ExaModel", True), ("ExaNLFIntegrator", False)):
        for key, m in methods(mock_text, cls).items():
            if key[0].startswith("~"):
                continue
            if not (m["ctor"] or m["virtual"] or m["override"] or nonvirtual_too):
                continue
            r = ref[cls].get(key)
            if r is None:
                bad.append(f"mock {cls}::{key[0]}({', '.join(key[1])}) is not declared by the reference class")
            elif not same(m, r) or (m["virtual"] or m["override"]) != (r["virtual"] or r["override"]):
                bad.append(f"mock {cls}::{key[0]}: {m} differs from the reference's {r}")
    # 2. what the reference overrides, the (real) MFEM base declares: the mock's base must declare the same virtuals
    base = methods(mock_text, "NonlinearFormIntegrator")
    for cls in ("ExaNLFIntegrator", "ICExaNLFIntegrator"):
        for key, r in ref[cls].items():
            if r["override"] and key[0] in ("AssemblePA", "AddMultPA", "AssembleGradPA", "AddMultGradPA", "AssembleGradDiagonalPA", "AssembleGradEA", "AssembleEA"):
                b = base.get(key)
                if b is None or not b["virtual"] or (b["ret"], b["const"]) != (r["ret"], r["const"]):
                    bad.append(f"mock mfem::NonlinearFormIntegrator lacks the virtual {key[0]}({', '.join(key[1])}) that {cls} overrides")
    # 3. every override of the adapters exists, identically, as a virtual of the reference base class
    for cls, refcls in (("HipExaModel", "ExaModel"), ("HipExaNLFIntegrator", "ExaNLFIntegrator"), ("HipExaModelLVec", "ExaModel"), ("HipExaNLFIntegratorLVec", "ExaNLFIntegrator")):
        for key, m in methods(adapt_text, cls).items():
            if not m["override"] or key[0].startswith("~"):
                continue
            r = ref[refcls].get(key)
            if r is None or not (r["virtual"] or r["override"]) or (m["ret"], m["const"]) != (r["ret"], r["const"]):
                bad.append(f"{cls}::{key[0]}({', '.join(key[1])}) overrides nothing in the reference's {refcls}")
    # 4. the pure virtuals of the reference's ExaModel are all implemented by HipExaModel
    have = methods(adapt_text, "HipExaModel")
    for key, r in ref["ExaModel"].items():
        if r["pure"] and key not in have:
            bad.append(f"HipExaModel does not implement the pure virtual ExaModel::{key[0]}")
    return bad


def _texts():
    return (open(MOCK).read(), open(ADAPT).read(), open(os.path.join(REF, "mechanics_model.hpp")).read(), open(os.path.join(REF, "mechanics_integrators.hpp")).read())


def test_reader_sees_the_reference_declarations():
    _, _, ref_model, ref_integ = _texts()
    em = methods(ref_model, "ExaModel")
    ms = em[("ModelSetup", ("const int", "const int", "const int", "const int", "const Vector &", "const Vector &", "const Vector &"))]
    assert ms["virtual"] and ms["pure"] and not ms["const"] and ms["ret"] == "void"
    dp = em[("calcDpMat", ("QuadratureFunction &",))]
    assert dp["pure"] and dp["const"]
    assert any(k[0] == "ExaModel" and len(k[1]) == 11 for k in em)          # the 11-argument constructor, mechanics_model.hpp:70-74
    ei = methods(ref_integ, "ExaNLFIntegrator")
    assert ei[("AddMultGradPA", ("const Vector &", "Vector &"))]["const"] and ei[("AddMultGradPA", ("const Vector &", "Vector &"))]["override"]
    assert ("AssembleEA", ("const FiniteElementSpace &", "Vector &")) in methods(ref_integ, "ICExaNLFIntegrator")
    assert ("AddMultGradPA", ("const Vector &", "Vector &")) not in methods(ref_integ, "ICExaNLFIntegrator")      # no PA gradient for B-bar


def test_mock_and_adapters_match_the_reference_headers():
    bad = mismatches(*_texts())
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("edit", [
    ("void AddMultPA(const mfem::Vector&, mfem::Vector&) const override {}", "void AddMultPA(const mfem::Vector&, mfem::Vector&) override {}"),       # const-ness
    ("const mfem::Vector& jacobian, const mfem::Vector& loc_grad,", "const mfem::Vector& jacobian, mfem::Vector& loc_grad,"),                          # a parameter type
    ("virtual void calcDpMat(mfem::QuadratureFunction& DpMat) const = 0;", "virtual void calcDpMat(mfem::QuadratureFunction& DpMat) const {}"),         # pure-ness
    ("virtual void AssembleEA(const FiniteElementSpace&, Vector&) {}", "virtual void AssembleEA(const FiniteElementSpace&, Vector&, int) {}"),          # the MFEM base
    ("int nStateVars, Assembly _assembly)\n      : numProps", "int nStateVars, Assembly _assembly, int extra)\n      : numProps"),                          # the constructor
])
def test_checker_notices_a_drifted_mock(edit):
    mock, adapt, ref_model, ref_integ = _texts()
    assert edit[0] in mock, "the edit's anchor is gone from the mock: update this test"
    assert mismatches(mock.replace(edit[0], edit[1], 1), adapt, ref_model, ref_integ)


def test_checker_notices_a_drifted_adapter():
    mock, adapt, ref_model, ref_integ = _texts()
    a = "void AddMultGradPA(const mfem::Vector& x, mfem::Vector& y) const override {"
    assert a in adapt
    assert mismatches(mock, adapt.replace(a, "void AddMultGradPA(const mfem::Vector& x, mfem::Vector& y) override {", 1), ref_model, ref_integ)
