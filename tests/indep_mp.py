"""Independent high-precision (mpmath, 50 digits) re-derivation of the reference factors in
MATRIX form (rotation matrices, no quaternion product formulas shared with the oracle), used to
cross-check the oracle's Jet-based residuals and Jacobians.  Central differences at 50 digits
give derivatives good to ~1e-25, so agreement to 1e-9 relative is a real check of the autodiff
restatement — including the d(q/|q|)/dq projector term of the normalising rotate."""
import mpmath as mp

mp.mp.dps = 50


def R_of(q):
    x, y, z, w = q
    n = mp.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return mp.matrix([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                      [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                      [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def vec(a):
    return mp.matrix([mp.mpf(float(v)) if not isinstance(v, mp.mpf) else v for v in a])


def apply(T, p):       # R p + t
    return R_of(T[:4]) * p + mp.matrix(T[4:7])


def apply_inv(T, p):   # R^T (p - t)
    return R_of(T[:4]).T * (p - mp.matrix(T[4:7]))


def pix(cam, pc):
    return mp.matrix([cam["fx"] * pc[0] / pc[2] + cam["cx"], cam["fy"] * pc[1] / pc[2] + cam["cy"]])


def mpcam(cam):
    return dict(fx=mp.mpf(cam["fx"]), fy=mp.mpf(cam["fy"]), cx=mp.mpf(cam["cx"]), cy=mp.mpf(cam["cy"]),
                e=[mp.mpf(float(v)) for v in cam["extrinsic"]])


def pose_only(x, ob, pw, cam0, w):
    pc = apply_inv(cam0["e"], apply_inv(x, pw))
    return w * (pix(cam0, pc) - ob)


def lift(ob, rho, cam):
    d = 1 / rho
    ps = mp.matrix([(ob[0] - cam["cx"]) / cam["fx"] * d, (ob[1] - cam["cy"]) / cam["fy"] * d, d])
    return apply(cam["e"], ps)


def two_frame(x, first_ob, ob, left, right, w):
    rho, T1, T2 = x[0], x[1:8], x[8:15]
    pw = apply(T1, lift(first_ob, rho, right))
    pc = apply_inv(left["e"], apply_inv(T2, pw))
    return w * (pix(left, pc) - ob)


def two_camera(x, left_ob, right_ob, left, right, w):
    pb = lift(right_ob, x[0], right)
    return w * (pix(left, apply_inv(left["e"], pb)) - left_ob)


def rot_zyx(yaw, pitch, roll):
    cz, sz, cy, sy, cx, sx = mp.cos(yaw), mp.sin(yaw), mp.cos(pitch), mp.sin(pitch), mp.cos(roll), mp.sin(roll)
    Rz = mp.matrix([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = mp.matrix([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = mp.matrix([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    return Rz * Ry * Rx


def lidar_plane(x3, mode, rpyxyz, Twc1, p, pa, n, w):
    r = list(rpyxyz)
    if mode == 0:
        r[1], r[2], r[5] = x3
    else:
        r[0], r[3], r[4] = x3
    lp = R_of(Twc1[:4]) * (rot_zyx(r[0], r[1], r[2]) * p + mp.matrix(r[3:6])) + mp.matrix(Twc1[4:7])
    d = lp - pa
    return mp.matrix([w * (d[0] * n[0] + d[1] * n[1] + d[2] * n[2])])


def fd_jacobian(f, x, h=mp.mpf(10) ** -22):
    x = [mp.mpf(v) for v in x]
    f0 = f(x)
    J = mp.zeros(len(f0), len(x))
    for k in range(len(x)):
        xp = list(x); xm = list(x)
        xp[k] += h; xm[k] -= h
        d = (f(xp) - f(xm)) / (2 * h)
        for i in range(len(f0)):
            J[i, k] = d[i]
    return f0, J
