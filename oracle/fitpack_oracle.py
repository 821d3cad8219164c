"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the smoothing-spline fit behind ``Scene.traj_to_spline`` (reference
``reconstruction/common.py:224-270``): ``scipy.interpolate.splprep(X, u=t, s=s, k=3)``, i.e. FITPACK's ``parcur`` with
``iopt = 0, ipar = 1, w = 1, ub = u[0], ue = u[-1], nest = m + 2k`` and its core ``fppara`` (P. Dierckx, "Curve and
Surface Fitting with Splines", 1993; FITPACK as vendored by scipy 1.15.3 -- a third-party dependency of the reference, not
under /root/reference).  The algorithm is restated routine by routine (``fpbspl``, ``fpgivs``, ``fprota``, ``fpback``,
``fpdisc``, ``fpknot``, ``fprati``, ``fppara``) with the published control flow: knots are added where the residual of the
least-squares spline is largest until f(p=inf) <= s, then the smoothing parameter p with F(p) = s is found by rational
interpolation, every linear system by row-wise Givens rotations.

Pinned by tests/test_traj_to_spline.py against ``scipy.interpolate.splprep`` itself (same knots, coefficients to rounding)
on seeded trajectories and on the golden fixture generated from the real reference's ``traj_to_spline``.

Plain Python loops: for test sizes (a few thousand samples) only.
"""
import numpy as np

TOL = 0.001          # parcur: tol
MAXIT = 20           # parcur: maxit


def fpbspl(t, k, x, l):
    """Non-zero B-splines of degree k at t[l-1] <= x < t[l] (1-based l as in FITPACK): h[0..k]."""
    h = np.zeros(k + 1)
    hh = np.zeros(k)
    h[0] = 1.0
    for j in range(1, k + 1):
        hh[:j] = h[:j]
        h[0] = 0.0
        for i in range(1, j + 1):
            li = l + i
            lj = li - j
            if t[li - 1] == t[lj - 1]:
                h[i] = 0.0
                continue
            f = hh[i - 1] / (t[li - 1] - t[lj - 1])
            h[i - 1] = h[i - 1] + f * (t[li - 1] - x)
            h[i] = f * (x - t[lj - 1])
    return h


def fpgivs(piv, ww):
    store = abs(piv)
    if store >= ww:
        dd = store * np.sqrt(1.0 + (ww / piv) ** 2)
    else:
        dd = ww * np.sqrt(1.0 + (piv / ww) ** 2)
    return ww / dd, piv / dd, dd          # cos, sin, new ww


def fprota(cos, sin, a, b):
    return cos * a - sin * b, cos * b + sin * a      # new a, new b


def fpback(a, z, n, k):
    """Back substitution with the upper triangular band a[n, k] (a[i, 0] the diagonal)."""
    c = np.zeros(n)
    k1 = k - 1
    c[n - 1] = z[n - 1] / a[n - 1, 0]
    i = n - 1
    for j in range(2, n + 1):
        store = z[i - 1]
        i1 = k1 if j > k1 else j - 1
        m = i
        for l in range(1, i1 + 1):
            m += 1
            store -= c[m - 1] * a[i - 1, l]
        c[i - 1] = store / a[i - 1, 0]
        i -= 1
    return c


def fpdisc(t, n, k2):
    """Discontinuity jumps of the k-th derivative of the B-splines at the interior knots: b[n - 2*k1, k2]."""
    k1 = k2 - 1
    k = k1 - 1
    nk1 = n - k1
    nrint = nk1 - k
    fac = nrint / (t[nk1] - t[k1 - 1])
    b = np.zeros((max(n - 2 * k1, 0), k2))
    h = np.zeros(2 * k1)
    for l in range(k2, nk1 + 1):
        lmk = l - k1
        for j in range(1, k1 + 1):
            ik = j + k1
            lj = l + j
            lk = lj - k2
            h[j - 1] = t[l - 1] - t[lk - 1]
            h[ik - 1] = t[l - 1] - t[lj - 1]
        lp = lmk
        for j in range(1, k2 + 1):
            jk = j
            prod = h[j - 1]
            for _ in range(k):
                jk += 1
                prod = prod * h[jk - 1] * fac
            lk = lp + k1
            b[lmk - 1, j - 1] = (t[lk - 1] - t[lp - 1]) / prod
            lp += 1
    return b


def fpknot(x, t, n, fpint, nrdata, nrint, istart=1):
    """Adds one knot in the interval with the largest residual that still holds data; returns (n + 1, nrint + 1)."""
    k = (n - nrint - 1) // 2
    fpmax = 0.0
    jbegin = istart
    number = maxpt = maxbeg = 0
    for j in range(1, nrint + 1):
        jpoint = nrdata[j - 1]
        if not (fpmax >= fpint[j - 1] or jpoint == 0):
            fpmax = fpint[j - 1]
            number = j
            maxpt = jpoint
            maxbeg = jbegin
        jbegin = jbegin + jpoint + 1
    ihalf = maxpt // 2 + 1
    nrx = maxbeg + ihalf
    nxt = number + 1
    if nxt <= nrint:
        for j in range(nxt, nrint + 1):
            jj = nxt + nrint - j
            fpint[jj] = fpint[jj - 1]
            nrdata[jj] = nrdata[jj - 1]
            jk = jj + k
            t[jk] = t[jk - 1]
    nrdata[number - 1] = ihalf - 1
    nrdata[nxt - 1] = maxpt - ihalf
    am = maxpt
    an = nrdata[number - 1]
    fpint[number - 1] = fpmax * an / am
    an = nrdata[nxt - 1]
    fpint[nxt - 1] = fpmax * an / am
    jk = nxt + k
    t[jk - 1] = x[nrx - 1]
    return n + 1, nrint + 1


def fprati(p1, f1, p2, f2, p3, f3):
    if p3 > 0.0:
        h1 = f1 * (f2 - f3)
        h2 = f2 * (f3 - f1)
        h3 = f3 * (f1 - f2)
        p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3)
    else:
        p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3)
    if f2 < 0.0:
        p3, f3 = p2, f2
    else:
        p1, f1 = p2, f2
    return p, p1, f1, p3, f3


def splprep(X, u, s, k=3):
    """``scipy.interpolate.splprep(X, u=u, s=s, k=k)`` for s > 0 restated.  X: (idim, m).  Returns ((t, [c_0..], k), info)
    with info = dict(fp, ier, n, p, knot_iterations, p_iterations)."""
    X = np.asarray(X, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    idim, m = X.shape
    assert s > 0 and m > k and np.all(u[1:] > u[:-1])
    k1, k2 = k + 1, k + 2
    nest = m + 2 * k
    ub, ue = u[0], u[-1]
    nmin = 2 * k1
    acc = TOL * s
    nmax = m + k1
    t = np.zeros(nest)
    fpint = np.zeros(nest)
    nrdata = np.zeros(nest, dtype=np.int64)
    n = nmin
    fpold = 0.0
    nplus = 0
    nrdata[0] = m - 2
    ier = 0
    fp0 = 0.0
    info = dict(knot_iterations=0, p_iterations=0, p=-1.0)
    q = np.zeros((m, k1))
    while True:                                           # fppara: do 200 iter = 1, m
        info['knot_iterations'] += 1
        if n == nmin:
            ier = -2
        nrint = n - nmin + 1
        nk1 = n - k1
        t[:k1] = ub
        t[n - k1:n] = ue
        # least-squares spline curve: observation rows rotated into the triangle a (band k1), right-hand sides z
        a = np.zeros((nk1, k1))
        z = np.zeros((idim, nk1))
        fp = 0.0
        l = k1
        for it in range(m):
            ui = u[it]
            xi = X[:, it].copy()
            while not (ui < t[l] or l == nk1):
                l += 1
            h = fpbspl(t, k, ui, l)
            q[it, :] = h
            j = l - k1
            for i in range(1, k1 + 1):
                j += 1
                piv = h[i - 1]
                if piv == 0.0:
                    continue
                cos, sin, a[j - 1, 0] = fpgivs(piv, a[j - 1, 0])
                for d in range(idim):
                    xi[d], z[d, j - 1] = fprota(cos, sin, xi[d], z[d, j - 1])
                if i == k1:
                    break
                i2 = 1
                for i1 in range(i + 1, k1 + 1):
                    i2 += 1
                    h[i1 - 1], a[j - 1, i2 - 1] = fprota(cos, sin, h[i1 - 1], a[j - 1, i2 - 1])
            fp += float(np.sum(xi * xi))
        if ier == -2:
            fp0 = fp
        c = np.array([fpback(a, z[d], nk1, k1) for d in range(idim)])
        fpms = fp - s
        if abs(fpms) < acc:
            break
        if fpms < 0.0:
            # ---- part 2: the smoothing spline, F(p) = s ----
            if ier == -2:
                break                                     # the least-squares polynomial is acceptable
            b = fpdisc(t, n, k2)
            p1, f1, p3, f3 = 0.0, fp0 - s, -1.0, fpms
            p = nk1 / float(np.sum(a[:, 0]))
            ich1 = ich3 = 0
            n8 = n - nmin
            done = False
            for iteration in range(1, MAXIT + 1):
                info['p_iterations'] += 1
                pinv = 1.0 / p
                cz = z.copy()
                g = np.zeros((nk1, k2))
                g[:, :k1] = a
                for it in range(1, n8 + 1):
                    h = np.zeros(k2 + 1)
                    h[:k2] = b[it - 1, :] * pinv
                    xi = np.zeros(idim)
                    for j in range(it, nk1 + 1):
                        piv = h[0]
                        cos, sin, g[j - 1, 0] = fpgivs(piv, g[j - 1, 0])
                        for d in range(idim):
                            xi[d], cz[d, j - 1] = fprota(cos, sin, xi[d], cz[d, j - 1])
                        if j == nk1:
                            break
                        i2 = k1 if j <= n8 else nk1 - j
                        for i in range(1, i2 + 1):
                            h[i], g[j - 1, i] = fprota(cos, sin, h[i], g[j - 1, i])
                            h[i - 1] = h[i]
                        h[i2] = 0.0
                c = np.array([fpback(g, cz[d], nk1, k2) for d in range(idim)])
                fp = 0.0
                l = k2
                for it in range(m):
                    if not (u[it] < t[l - 1] or l > nk1):
                        l += 1
                    l0 = l - k2
                    term = 0.0
                    for d in range(idim):
                        fac = float(np.dot(c[d, l0:l0 + k1], q[it]))
                        term += (fac - X[d, it]) ** 2
                    fp += term
                fpms = fp - s
                if abs(fpms) < acc:
                    done = True
                    break
                if iteration == MAXIT:
                    ier = 3
                    break
                p2, f2 = p, fpms
                if ich3 == 0:
                    if f2 - f3 <= acc:                    # the initial choice of p is too large
                        p3, f3 = p2, f2
                        p = p * 0.04
                        if p <= p1:
                            p = p1 * 0.9 + p2 * 0.1
                        continue
                    if f2 < 0.0:
                        ich3 = 1
                if ich1 == 0:
                    if f1 - f2 <= acc:                    # the initial choice of p is too small
                        p1, f1 = p2, f2
                        p = p / 0.04
                        if p3 < 0.0:
                            continue
                        if p >= p3:
                            p = p2 * 0.1 + p3 * 0.9
                        continue
                    if f2 > 0.0:
                        ich1 = 1
                if f2 >= f1 or f2 <= f3:
                    ier = 2
                    break
                p, p1, f1, p3, f3 = fprati(p1, f1, p2, f2, p3, f3)
            if done or ier > 0:
                ier = ier if ier > 0 else 0
            info['p'] = p
            break
        if n == nmax:
            ier = -1
            break
        if n == nest:
            ier = 1
            break
        # ---- more knots ----
        if ier == 0:
            npl1 = nplus * 2
            rn = float(nplus)
            if fpold - fp > acc:
                npl1 = int(rn * fpms / (fpold - fp))
            nplus = min(nplus * 2, max(npl1, nplus // 2, 1))
        else:
            nplus = 1
            ier = 0
        fpold = fp
        # residual of every knot interval; a sample sitting on a knot is shared half / half
        fpart = 0.0
        i = 1
        l = k2
        new = 0
        for it in range(m):
            if not (u[it] < t[l - 1] or l > nk1):
                new = 1
                l += 1
            l0 = l - k2
            term = 0.0
            for d in range(idim):
                fac = float(np.dot(c[d, l0:l0 + k1], q[it]))
                term += (fac - X[d, it]) ** 2
            if new:
                store = term * 0.5
                fpint[i - 1] = fpart + store
                i += 1
                fpart = store
                new = 0
            else:
                fpart += term
        fpint[nrint - 1] = fpart
        for _ in range(nplus):
            n, nrint = fpknot(u, t, n, fpint, nrdata, nrint, 1)
            if n == nmax or n == nest:
                break
        if n == nmax:                                     # fppara label 10: the knots of the interpolating spline (k odd: the data sites)
            mk1 = m - k1
            k3 = k // 2
            i, j = k2, k3 + 2
            for _ in range(mk1):
                t[i - 1] = u[j - 1] if k3 * 2 != k else (u[j - 1] + u[j - 2]) * 0.5
                i += 1
                j += 1
    info.update(fp=fp, ier=ier, n=n)
    nk1 = n - k1
    return (t[:n].copy(), [c[d, :nk1].copy() for d in range(idim)], k), info
