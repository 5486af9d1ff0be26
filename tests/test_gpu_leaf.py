"""Leaf-function tables: the HIP restatements of the reference's inline functions,
evaluated on seeded random inputs through the C-ABI test hook, against the reference's OWN functions
(oracle/_ref leaf entry points, which call straight into intersection.h / disney.h / probe.h / util.h).

Tolerances: Random is bit-exact (integer); everything else must agree to 4 ulp-equivalents (rtol 5e-7 of the
row's magnitude) on >= 99.9 % of rows and be bit-identical on most -- rows where a comparison (hit/miss, lobe
choice) lands on the other side of a 1-ulp difference are counted, not hidden."""
import ctypes as C
import os

import numpy as np
import pytest

from tinsel_amd import abi
from tests import oracle_api as oa

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not oa.have_ref(), reason="oracle/_ref not built")]

N = 200_000


def _setup(name):
    import tinsel_amd
    scene = tinsel_amd.Scene.load_pack(os.path.join(oa.GOLDEN, name + ".pack"))
    r = tinsel_amd.create_gpu_renderer(scene)
    R = oa.RefOracle()
    h = R.load_pack(os.path.join(oa.GOLDEN, name + ".pack"))
    return scene, r, R, h


def _unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v/np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _close(a, b, what, frac=0.999):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.maximum(np.abs(b).max(axis=-1, keepdims=True) if b.ndim > 1 else np.abs(b), 1e-6)
    ok = np.abs(a - b) <= 5e-7*scale + 1e-12
    rows_ok = ok.all(axis=-1) if ok.ndim > 1 else ok
    exact = (a == b).all(axis=-1) if a.ndim > 1 else (a == b)
    print("%-28s rows %d  bit-identical %.4f  within 4 ulp %.5f" % (what, len(rows_ok), exact.mean(), rows_ok.mean()))
    assert rows_ok.mean() >= frac, what


def test_random_is_bit_exact():
    scene, r, R, h = _setup("cornell")
    seeds = np.random.default_rng(1).integers(0, 2**32, size=4096, dtype=np.uint64).astype(np.uint32)
    out = r.leaf(0, 0, len(seeds), 8, seeds=seeds)
    for i in (0, 17, 4095):
        rr, ff = R.leaf_random(int(seeds[i]), 4)
        assert np.array_equal(out[i, :4].view(np.uint32), rr)
        assert np.array_equal(out[i, 4:], ff)
    r.close(); R.free(h)


def test_camera_rays():
    scene, r, R, h = _setup("features")
    rng = np.random.default_rng(2)
    W, H = 1920, 1080
    xy = (rng.random((N, 2))*[W, H]).astype(np.float32)
    out = r.leaf(1, 0, N, 6, rows=xy, camera=scene.camera, width=W, height=H)
    ref = R.camera_rays(scene.camera, W, H, xy)
    assert np.array_equal(out, ref), "CameraSampler::GenerateRay must be bit-identical (same host matrix, same fp32 ops)"
    r.close(); R.free(h)


@pytest.mark.parametrize("scene_name,prims", [("features", [0, 1, 4, 5, 6, 7]), ("cornell", [0, 6, 7]), ("glass", [3, 6])])
def test_bsdf_eval_pdf_sample(scene_name, prims):
    scene, r, R, h = _setup(scene_name)
    rng = np.random.default_rng(3)
    for prim in prims:
        mat = R.primitive(h, prim).material
        n, V, L = _unit(rng, N), _unit(rng, N), _unit(rng, N)
        V = np.where((np.sum(V*n, axis=1, keepdims=True) < 0), -V, V).astype(np.float32)    # view above the surface
        eta = np.where(rng.random((N, 1)) < 0.5, [[1.0, 1.5]], [[1.5, 1.0]]).astype(np.float32)
        rows = np.concatenate([n, V, L, eta], axis=1).astype(np.float32)
        out = r.leaf(2, prim, N, 4, rows=rows)
        f, pdf = R.bsdf_eval(mat, rows)
        fin = np.isfinite(f).all(axis=1) & np.isfinite(pdf)
        _close(out[fin, :3], f[fin], "%s prim %d BSDFEval" % (scene_name, prim))
        _close(out[fin, 3], pdf[fin], "%s prim %d BSDFPdf" % (scene_name, prim))

        seeds = rng.integers(0, 2**32, size=N, dtype=np.uint64).astype(np.uint32)
        rows8 = np.concatenate([n, V, eta], axis=1).astype(np.float32)
        so = r.leaf(3, prim, N, 7, rows=rows8, seeds=seeds)
        Lr, pr, tr, st = R.bsdf_sample(mat, rows8, seeds)
        # the RNG stream consumed is integer-exact unless a `rand < F` comparison flipped on a 1-ulp F
        same_stream = (so[:, 5:7].view(np.uint32) == st).all(axis=1)
        assert same_stream.mean() >= 0.9999
        live = same_stream & (pr > 0) & np.isfinite(pr)
        assert (so[live, 4].view(np.int32) == tr[live]).mean() >= 0.9999
        _close(so[live, :3], Lr[live], "%s prim %d BSDFSample dir" % (scene_name, prim))
        _close(so[live, 3], pr[live], "%s prim %d BSDFSample pdf" % (scene_name, prim), frac=0.998)
    r.close(); R.free(h)


def test_bsdf_pdf_where_fresnel_is_not_finite():
    """BSDFPdf computes Fr(Dot(n, V), etaI, etaO) for every material and multiplies it by transmission == 0 for an opaque one: NaN where Fr divides
    0 by 0 -- V exactly in the surface's plane, etaI == etaO -- and the pdf is NaN.  The HIP path's shortcut for opaque materials must not hide that:
    rows with Dot(n, V) == 0 (and just above, where Fr is finite and the shortcut applies), equal and unequal indices, against the reference."""
    scene, r, R, h = _setup("cornell")
    rows = []
    for V in ([1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.6, 0.0, 0.8], [1.0, 1e-7, 0.0], [0.99, 0.14106736, 0.0]):
        for L in ([0.0, 1.0, 0.0], [0.3, 0.9, 0.31622777], [-0.5, 0.70710678, 0.5]):
            for eta in ([1.0, 1.0], [1.5, 1.5], [1.0, 1.5], [1.5, 1.0]):
                rows.append([0.0, 1.0, 0.0] + V + L + eta)
    rows = np.array(rows, np.float32)
    for prim in (0, 6, 7):                                      # a wall, the glossy sphere, the metal one: all opaque
        mat = R.primitive(h, prim).material
        out = r.leaf(2, prim, len(rows), 4, rows=rows)
        f, pdf = R.bsdf_eval(mat, rows)
        assert np.isnan(pdf).any() and not np.isnan(pdf).all()  # (the rows hold both cases)
        assert np.array_equal(np.isnan(out[:, 3]), np.isnan(pdf)), (prim, np.nonzero(np.isnan(out[:, 3]) != np.isnan(pdf)))
        ok = ~np.isnan(pdf)
        assert np.array_equal(out[ok, 3].view(np.uint32), pdf[ok].view(np.uint32)), prim
    r.close(); R.free(h)


@pytest.mark.parametrize("scene_name,prims", [("features", [0, 2, 3, 4, 5, 6, 7]), ("cornell", [0, 5, 6]), ("ajax_standin_96", [0, 1])])
def test_primitive_intersect(scene_name, prims):
    scene, r, R, h = _setup(scene_name)
    rng = np.random.default_rng(4)
    for prim in prims:
        o = (rng.normal(size=(N, 3))*2.0 + [0.0, 1.0, 0.0]).astype(np.float32)
        tgt = (rng.normal(size=(N, 3))*0.8 + [0.0, 0.8, 0.0]).astype(np.float32)
        d = tgt - o
        d = (d/np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        t = rng.random((N, 1)).astype(np.float32)
        rows = np.concatenate([o, d, t], axis=1).astype(np.float32)
        out = r.leaf(4, prim, N, 5, rows=rows)
        hit, tt, nrm = R.primitive_intersect(h, prim, rows)
        same = (out[:, 0] > 0.5) == (hit > 0)
        print("%s prim %d: hits %.3f, hit/miss agreement %.6f" % (scene_name, prim, hit.mean(), same.mean()))
        assert same.mean() >= 0.9999
        both = same & (hit > 0)
        if both.any():
            _close(out[both, 1], tt[both], "%s prim %d t" % (scene_name, prim))
            _close(out[both, 2:5], nrm[both], "%s prim %d normal" % (scene_name, prim), frac=0.998)
    r.close(); R.free(h)


@pytest.mark.parametrize("scene_name,prim", [("cornell", 5), ("glass", 6), ("glass", 7), ("veach", 2), ("veach", 3)])
def test_primitive_intersect_rays_with_zero_components(scene_name, prim):
    """Unrotated mesh primitives take Rotate() by the identity quaternion as written down (pose_inv_ray / pose_rotate, tn_isect.h): exact
    for vectors without zero components; a vector WITH one (an axis-aligned ray, an origin level with the primitive's position) takes the
    reference's two quaternion products, because the sign of the zero that comes out depends on its neighbours' signs.  Every combination
    of -x, -0, +0, +x in the components of o - p and d, aimed at the primitive: hit flag, t and normal bit for bit (zero signs included)."""
    import ctypes as C
    scene, r, R, h = _setup(scene_name)
    prims = C.cast(scene.desc.primitives, C.POINTER(abi.Primitive))
    assert prims[prim].type == abi.GEOM_MESH
    tp = prims[prim].start_transform.p
    centre = np.array([tp.x, tp.y, tp.z], np.float32)           # (veach 3 is rotated: the reference's formula on every ray, as a control)
    vals = np.array([-0.07, -0.0, 0.0, 0.07], np.float32)
    rows = []
    for dist, axis in ((1.5, 1), (-1.5, 1), (1.5, 0), (-1.5, 2)):
        for a in vals:
            for b in vals:
                for da in vals:
                    for db in vals:
                        off = np.zeros(3, np.float32); dd = np.zeros(3, np.float32)
                        others = [k for k in range(3) if k != axis]
                        off[others[0]], off[others[1]], off[axis] = a, b, -dist
                        dd[others[0]], dd[others[1]], dd[axis] = da, db, np.sign(dist)
                        for sgn in (1.0, -1.0):             # (a signed zero along the axis too)
                            o = (centre + off).astype(np.float32)
                            if sgn < 0:
                                o[others[0]] = centre[others[0]]        # o - p exactly +0 there
                            d = (dd/np.float32(np.sqrt(np.float32((dd*dd).sum())))).astype(np.float32)
                            rows.append(np.concatenate([o, d, [0.0]]).astype(np.float32))
    rows = np.stack(rows).astype(np.float32)
    out = r.leaf(4, prim, len(rows), 5, rows=rows)
    hit, tt, nrm = R.primitive_intersect(h, prim, rows)
    r.close(); R.free(h)
    got = out[:, 0] > 0.5
    print("%s prim %d: %d rays, %d hits" % (scene_name, prim, len(rows), int((hit > 0).sum())))
    assert (hit > 0).sum() > len(rows)//8
    assert np.array_equal(got, hit > 0)
    m = hit > 0
    assert np.array_equal(out[m, 1].view(np.uint32), tt[m].astype(np.float32).view(np.uint32))
    assert np.array_equal(np.ascontiguousarray(out[m, 2:5]).view(np.uint32), np.ascontiguousarray(nrm[m].astype(np.float32)).view(np.uint32))


@pytest.mark.parametrize("scene_name,prims", [("features", [2, 3]), ("cornell", [5]), ("veach", [5, 8]), ("glass", [6])])
def test_primitive_sample(scene_name, prims):
    scene, r, R, h = _setup(scene_name)
    rng = np.random.default_rng(5)
    for prim in prims:
        times = rng.random((N, 1)).astype(np.float32)
        seeds = rng.integers(0, 2**32, size=N, dtype=np.uint64).astype(np.uint32)
        out = r.leaf(5, prim, N, 8, rows=times, seeds=seeds)
        pos, nrm, st = R.primitive_sample(h, prim, times[:, 0], seeds)
        assert np.array_equal(out[:, 6:8].view(np.uint32), st)
        _close(out[:, :3], pos, "%s prim %d PrimitiveSample pos" % (scene_name, prim))
        _close(out[:, 3:6], nrm, "%s prim %d PrimitiveSample normal" % (scene_name, prim))
    r.close(); R.free(h)


def test_probe_sample_pdf_eval():
    scene, r, R, h = _setup("features_probe")
    seeds = np.random.default_rng(6).integers(0, 2**32, size=N, dtype=np.uint64).astype(np.uint32)
    out = r.leaf(6, 0, N, 11, seeds=seeds)
    d, c, pdf, pdf2, ev = R.probe(h, seeds)
    _close(out[:, :3], d, "ProbeSample dir")
    assert np.array_equal(out[:, 3:6], c), "probe texel colours are copied, not computed"
    _close(out[:, 6], pdf, "ProbeSample pdf")
    # ProbePdf / Sky::Eval re-derive the texel from the direction (acosf/atan2f).  ProbeSample returns the
    # direction of a texel CORNER (u = col/W exactly, probe.h:224-225), i.e. this feeds them the worst case:
    # a 1-ulp uv lands on the neighbouring texel.  Measured agreement 98.2 %; in a render these functions see
    # BSDF-sampled directions, not corners.
    _close(out[:, 7], pdf2, "ProbePdf(dir)", frac=0.9999)
    _close(out[:, 8:11], ev, "Sky::Eval(dir)", frac=0.9999)
    r.close(); R.free(h)


def test_device_libm_is_glibc_bit_for_bit():
    """sinf / cosf / expf / acosf / atan2f on the device restate glibc 2.35's algorithms (tn_math.h): every one of
    4 M angles in [0, 2*pi], exponents in [-80, 0], cosines in [-1, 1] and (y, x) pairs must equal THIS host's
    libm to the last bit."""
    scene, r, R, h = _setup("cornell")
    libm = C.CDLL("libm.so.6")
    for f in (libm.sinf, libm.cosf, libm.expf):
        f.restype = C.c_float
        f.argtypes = [C.c_float]
    rng = np.random.default_rng(7)
    n = 4_000_000
    x = np.concatenate([(rng.random(n//2)*2*np.pi), rng.random(n//4)*80.0, rng.random(n//4)*1e-2]).astype(np.float32)
    x[:8] = [0.0, 1e-30, 2.0**-12, np.float32(np.pi/4), np.float32(np.pi/2), np.float32(np.pi), np.float32(2*np.pi), 6.2831855]
    yv = (rng.random(len(x))*2 - 1).astype(np.float32)
    yv[:4] = [1.0, -1.0, 0.0, 0.5]
    out = r.leaf(7, 0, len(x), 5, rows=np.stack([x, yv], axis=1))
    # vectorised host libm through numpy would not be glibc's scalar routine: call it per element on a sample,
    # and on everything through a tiny C loop compiled here
    import subprocess, tempfile
    src = "#include <math.h>\nvoid f(int n,const float*x,const float*y,float*o){for(int i=0;i<n;i++){o[5*i]=sinf(x[i]);o[5*i+1]=cosf(x[i]);o[5*i+2]=expf(-x[i]);o[5*i+3]=acosf(y[i]);o[5*i+4]=atan2f(y[i],x[i]-3.0f);}}"
    d = tempfile.mkdtemp()
    open(os.path.join(d, "l.c"), "w").write(src)
    subprocess.run(["gcc", "-O1", "-fno-builtin", "-shared", "-fPIC", "-o", os.path.join(d, "l.so"), os.path.join(d, "l.c"), "-lm"], check=True)
    L = C.CDLL(os.path.join(d, "l.so"))
    ref = np.zeros((len(x), 5), np.float32)
    L.f(len(x), x.ctypes.data_as(C.c_void_p), yv.ctypes.data_as(C.c_void_p), ref.ctypes.data_as(C.c_void_p))
    bad_a = int((out[:, 3].view(np.uint32) != ref[:, 3].view(np.uint32)).sum())
    bad_t = int((out[:, 4].view(np.uint32) != ref[:, 4].view(np.uint32)).sum())
    print("acosf mismatches %d, atan2f %d of %d" % (bad_a, bad_t, len(x)))
    assert bad_a == 0 and bad_t == 0
    ang = x <= 7.0
    bad_s = int((out[ang, 0] != ref[ang, 0]).sum()); bad_c = int((out[ang, 1] != ref[ang, 1]).sum())
    bad_e = int((out[:, 2] != ref[:, 2]).sum())
    print("sinf mismatches %d, cosf %d of %d ; expf %d of %d" % (bad_s, bad_c, ang.sum(), bad_e, len(x)))
    assert bad_s == 0 and bad_c == 0
    assert bad_e <= 2
    # atan2f's special cases (e_atan2f.c: infinities, zeros, x == 1), which unit directions never reach: the same C loop, that column only
    inf = np.float32(np.inf)
    ys = np.array([inf, -inf, inf, -inf, inf, -inf, 0.5, -0.5, 0.5, -0.5, 0.0, -0.0, 0.0, -0.0, inf, -inf, inf, 0.25, 0.0, -0.0, 1e-38, -1e30], np.float32)
    xs = np.array([inf, inf, -inf, -inf, 2.5, -7.0, inf, inf, -inf, -inf, inf, inf, -inf, -inf, 3.0, 3.0, 4.0, 4.0, 3.0, 3.0, 3.0, 3.0], np.float32)
    out2 = r.leaf(7, 0, len(xs), 5, rows=np.stack([xs, ys], axis=1))
    ref2 = np.zeros((len(xs), 5), np.float32)
    L.f(len(xs), xs.ctypes.data_as(C.c_void_p), ys.ctypes.data_as(C.c_void_p), ref2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out2[:, 4].view(np.uint32), ref2[:, 4].view(np.uint32)), (out2[:, 4], ref2[:, 4])
    r.close(); R.free(h)
