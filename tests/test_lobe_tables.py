"""Oracle BxDF lobes against INDEPENDENT value tables (tests/golden/lobe_tables.json).

The tables were computed with mpmath (50 digits) by tools/make_lobe_tables.py from formulas transcribed
straight from the reference's src/bxdf.rs / src/microfacet.rs / src/util.rs -- not from the oracle and not from
the device headers -- so these tests tie the oracle's f / pdf / sample_f of every lobe on the hot path
(LambertianReflection, MicrofacetReflection with conductor and dielectric Fresnel, MicrofacetTransmission,
FresnelSpecular, SpecularReflection) to the reference's own text.  Tolerance: 1e-9 relative (+1e-13 absolute);
both sides evaluate the same real-valued expression, the oracle in binary64 with <= 2 ulp elementary functions.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests import oracle_ffi as O

REL, ABS = 1e-9, 1e-13
PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lobe_tables.json")


class oracle_lobe(C.Structure):
    _fields_ = [("kind", C.c_int32), ("fresnel", C.c_int32), ("color", C.c_double * 3), ("t", C.c_double * 3),
                ("eta_i", C.c_double), ("eta_t", C.c_double), ("eta", C.c_double * 3), ("k", C.c_double * 3),
                ("alpha_x", C.c_double), ("alpha_y", C.c_double), ("eta_a", C.c_double), ("eta_b", C.c_double)]


@pytest.fixture(scope="module")
def tables():
    with open(PATH) as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def L():
    lib = O.lib()
    dp = C.POINTER(C.c_double)
    lib.oracle_lobe_eval.argtypes = [C.POINTER(oracle_lobe), dp, dp, dp, dp]
    lib.oracle_lobe_sample.argtypes = [C.POINTER(oracle_lobe), dp, C.c_double, C.c_double, C.POINTER(C.c_uint64), dp, dp, dp]
    return lib


def close(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.all(np.abs(a - b) <= ABS + REL * np.abs(b))


def make_lobe(d):
    l = oracle_lobe()
    l.kind, l.fresnel = d["kind"], d["fresnel"]
    l.color[:], l.t[:], l.eta[:], l.k[:] = d["color"], d["t"], d["eta"], d["k"]
    l.eta_i, l.eta_t, l.alpha_x, l.alpha_y, l.eta_a, l.eta_b = (d["eta_i"], d["eta_t"], d["alpha_x"], d["alpha_y"],
                                                               d["eta_a"], d["eta_b"])
    return l


LOBES = ["lambert", "microfacet_conductor_iso", "microfacet_conductor_aniso", "microfacet_dielectric_plastic",
         "microfacet_transmission", "fresnel_specular", "specular_reflection_mirror", "specular_reflection_dielectric"]


@pytest.mark.parametrize("name", LOBES)
def test_lobe_f_and_pdf_tables(L, tables, name):
    entry = next(e for e in tables["lobes"] if e["name"] == name)
    lobe = make_lobe(entry["lobe"])
    assert len(entry["eval"]) >= 20
    nonzero = 0
    for row in entry["eval"]:
        f = (C.c_double * 3)()
        pdf = C.c_double()
        L.oracle_lobe_eval(C.byref(lobe), O.vec(*row["wo"]), O.vec(*row["wi"]), f, C.byref(pdf))
        assert close(f[:], row["f"]), (name, row, f[:])
        assert close(pdf.value, row["pdf"]), (name, row, pdf.value)
        nonzero += any(v != 0.0 for v in row["f"]) or row["pdf"] != 0.0
    if name not in ("fresnel_specular", "specular_reflection_mirror", "specular_reflection_dielectric"):
        assert nonzero >= 4  # the table is not a list of zeros


@pytest.mark.parametrize("name", LOBES)
def test_lobe_sample_f_tables(L, tables, name):
    entry = next(e for e in tables["lobes"] if e["name"] == name)
    lobe = make_lobe(entry["lobe"])
    assert len(entry["sample"]) >= 20
    nonzero = 0
    for row in entry["sample"]:
        f, wi = (C.c_double * 3)(), (C.c_double * 3)()
        pdf = C.c_double()
        key = (C.c_uint64 * 3)(*row["rng_key"])
        L.oracle_lobe_sample(C.byref(lobe), O.vec(*row["wo"]), row["u"][0], row["u"][1], key, f, wi, C.byref(pdf))
        assert close(wi[:], row["wi"]), (name, row, wi[:])
        assert close(pdf.value, row["pdf"]), (name, row, pdf.value)
        assert close(f[:], row["f"]), (name, row, f[:])
        nonzero += row["pdf"] != 0.0
    assert nonzero >= 12


def test_rng_draws_feeding_default_sample_f(L, tables):
    """The two entropy draws of default_sample_f come from the ABI's counter stream (rt_abi.h): the table's
    r1, r2 were computed from the spec with Python integers."""
    entry = next(e for e in tables["lobes"] if e["name"] == "lambert")
    for row in entry["sample"]:
        out = (C.c_double * 2)()
        L.oracle_rng_draws(row["rng_key"][0], row["rng_key"][1], row["rng_key"][2], 2, out)
        assert out[0] == row["r"][0] and out[1] == row["r"][1]


def test_scalar_tables(L, tables):
    s = tables["scalars"]
    assert len(s["fr_dielectric"]) >= 20 and len(s["fr_conductor"]) >= 20 and len(s["tr"]) >= 20
    for r in s["fr_dielectric"]:
        assert close(L.oracle_fr_dielectric(r["cos"], r["eta_i"], r["eta_t"]), r["value"]), r
    for r in s["fr_conductor"]:
        out = (C.c_double * 3)()
        L.oracle_fr_conductor(r["cos"], O.vec(*r["eta"]), O.vec(*r["k"]), out)
        assert close(out[:], r["value"]), (r, out[:])
    for r in s["tr"]:
        ax, ay = r["alpha"]
        wo, wh = O.vec(*r["wo"]), O.vec(*r["wh"])
        assert close(L.oracle_tr_d(ax, ay, wh), r["d"]), r
        assert close(L.oracle_tr_lambda(ax, ay, wo), r["lambda"]), r
        assert close(L.oracle_tr_g(ax, ay, wo, wh), r["g"]), r
        assert close(L.oracle_tr_pdf(ax, ay, wo, wh), r["pdf"]), r
        out = (C.c_double * 3)()
        L.oracle_tr_sample_wh(ax, ay, wo, r["u"][0], r["u"][1], out)
        assert close(out[:], r["sample_wh"]), (r, out[:])
    for r in s["roughness_to_alpha"]:
        assert close(L.oracle_tr_roughness_to_alpha(r["roughness"]), r["alpha"]), r
