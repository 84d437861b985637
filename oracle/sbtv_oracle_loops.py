"""Second, independent restatement of the innermost reference functions as scalar loops with MATLAB's 1-based
indexing emulated literally.  TEST INFRASTRUCTURE ONLY (same rules as sbtv_oracle.py; pure-Python loops, so for
tiny cases only).  Its single purpose: pin the vectorised NumPy oracle against slips of vectorisation (slicing
off-by-one, axis mix-ups, wrap direction) by reading the same MATLAB lines a second time in the most literal form.
PARITY STATUS: unpinned against MATLAB output, like the main oracle (no MATLAB/Octave in the image).

Every array here is a dict-free list of lists A[i][j] addressed through helpers that take 1-based (i, j).
"""
from __future__ import annotations

import math


class M1:
    """A MATLAB matrix with 1-based element access."""

    def __init__(self, rows, cols, fill=0.0):
        self.m, self.n = rows, cols
        self.a = [[fill] * cols for _ in range(rows)]

    @classmethod
    def of(cls, arr):
        out = cls(len(arr), len(arr[0]))
        for i in range(out.m):
            for j in range(out.n):
                out.a[i][j] = float(arr[i][j])
        return out

    def g(self, i, j):
        assert 1 <= i <= self.m and 1 <= j <= self.n, (i, j, self.m, self.n)
        return self.a[i - 1][j - 1]

    def s(self, i, j, v):
        assert 1 <= i <= self.m and 1 <= j <= self.n, (i, j, self.m, self.n)
        self.a[i - 1][j - 1] = v

    def tolist(self):
        return [row[:] for row in self.a]


def DivergenceIm(p1, p2):
    """utils/chambolle_prox_TV_stop.m:152-159, element by element."""
    m, n = p1.m, p1.n
    divp = M1(m, n)
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            # v = [p2(:,1)  p2(:,2:end-1)-p2(:,1:end-2)  -p2(:,end)]          (:153-154)
            if j == 1:
                v = p2.g(i, 1)
            elif j == n:
                v = -p2.g(i, n)
            else:
                v = p2.g(i, j) - p2.g(i, j - 1)       # column j of z is p2(:,j) - p2(:,j-1), j = 2..n-1
            # u = [p1(1,:); p1(2:end-1,:)-p1(1:end-2,:); -p1(end,:)]            (:156-157)
            if i == 1:
                u = p1.g(1, j)
            elif i == m:
                u = -p1.g(m, j)
            else:
                u = p1.g(i, j) - p1.g(i - 1, j)
            divp.s(i, j, v + u)                                                 # :159
    return divp


def GradientIm(u):
    """utils/chambolle_prox_TV_stop.m:161-166."""
    m, n = u.m, u.n
    dux, duy = M1(m, n), M1(m, n)
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            dux.s(i, j, u.g(i + 1, j) - u.g(i, j) if i < m else 0.0)            # [z; zeros(1,n)]
            duy.s(i, j, u.g(i, j + 1) - u.g(i, j) if j < n else 0.0)            # [z zeros(m,1)]
    return dux, duy


def chambolle_prox_TV_stop(g, lam, MaxIter, tol=1e-3, tau=0.249):
    """utils/chambolle_prox_TV_stop.m:68-149, zero dual start; returns f, px, py, k, err."""
    m, n = g.m, g.n
    px, py = M1(m, n), M1(m, n)                                                 # :68-69
    cont, k, err = True, 0, 0.0
    while cont:                                                                 # :117
        k += 1
        divp = DivergenceIm(px, py)                                             # :120
        u = M1(m, n)
        for i in range(1, m + 1):
            for j in range(1, n + 1):
                u.s(i, j, divp.g(i, j) - g.g(i, j) / lam)                       # :121
        upx, upy = GradientIm(u)                                                # :123
        acc = 0.0
        npx, npy = M1(m, n), M1(m, n)
        # err sums over x(:) (column-major order) with the OLD px, py            (:125, quirk Q4)
        for j in range(1, n + 1):
            for i in range(1, m + 1):
                tmp = math.sqrt(upx.g(i, j) ** 2 + upy.g(i, j) ** 2)            # :124
                acc += (-upx.g(i, j) + tmp * px.g(i, j)) ** 2 + (-upy.g(i, j) + tmp * py.g(i, j)) ** 2
                npx.s(i, j, (px.g(i, j) + tau * upx.g(i, j)) / (1 + tau * tmp))  # :126
                npy.s(i, j, (py.g(i, j) + tau * upy.g(i, j)) / (1 + tau * tmp))  # :127
        err = acc ** 0.5
        px, py = npx, npy
        cont = (k < MaxIter) and (err > tol)                                    # :128
    divp = DivergenceIm(px, py)
    f = M1(m, n)
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            f.s(i, j, g.g(i, j) - lam * divp.g(i, j))                           # :149
    return f, px, py, k, err


def wraparound(x, mm, nm):
    """SALSA/conv2c.m:7-50 for a mask of size mm x nm, copying the blocks exactly as written."""
    mx, nx = x.m, x.n
    if mm > mx or nm > nx:
        raise ValueError("Mask does not fit inside array")
    mo, no = (1 + mm) // 2, (1 + nm) // 2
    ml, nl = mo - 1, no - 1
    mr, nr = mm - mo, nm - no
    me, ne = mx - ml + 1, nx - nl + 1
    mt, nt = mx + ml, nx + nl
    my, ny = mx + mm - 1, nx + nm - 1
    y = M1(my, ny)

    def copy(r0, r1, c0, c1, sr0, sc0):           # y(r0:r1, c0:c1) = x(sr0:.., sc0:..)
        for di in range(r1 - r0 + 1):
            for dj in range(c1 - c0 + 1):
                y.s(r0 + di, c0 + dj, x.g(sr0 + di, sc0 + dj))
    copy(mo, mt, no, nt, 1, 1)
    if ml > 0:
        copy(1, ml, no, nt, me, 1)
        if nl > 0:
            copy(1, ml, 1, nl, me, ne)
        if nr > 0:
            copy(1, ml, nt + 1, ny, me, 1)
    if mr > 0:
        copy(mt + 1, my, no, nt, 1, 1)
        if nl > 0:
            copy(mt + 1, my, 1, nl, 1, ne)
        if nr > 0:
            copy(mt + 1, my, nt + 1, ny, 1, 1)
    if nl > 0:
        copy(mo, mt, 1, nl, 1, ne)
    if nr > 0:
        copy(mo, mt, nt + 1, ny, 1, 1)
    return y


def conv2_valid(x, h):
    """MATLAB conv2(x, h, 'valid'): c(i,j) = sum_{p,q} h(p,q) x(i+mm-p, j+nm-q) (true convolution: flipped mask)."""
    mm, nm = h.m, h.n
    out = M1(x.m - mm + 1, x.n - nm + 1)
    for i in range(1, out.m + 1):
        for j in range(1, out.n + 1):
            acc = 0.0
            for p in range(1, mm + 1):
                for q in range(1, nm + 1):
                    acc += h.g(p, q) * x.g(i + mm - p, j + nm - q)
            out.s(i, j, acc)
    return out


def conv2c(x, h):
    """SALSA/conv2c.m:1-4."""
    return conv2_valid(wraparound(x, h.m, h.n), h)


def TVnorm(x):
    """utils/TVnorm.m:2 with SALSA/diffh.m (h = [0 1 -1]) and SALSA/diffv.m (h = [0 1 -1]')."""
    dh = conv2c(x, M1.of([[0, 1, -1]]))
    dv = conv2c(x, M1.of([[0], [1], [-1]]))
    acc = 0.0
    for i in range(1, x.m + 1):
        for j in range(1, x.n + 1):
            acc += math.sqrt(dh.g(i, j) ** 2 + dv.g(i, j) ** 2)
    return acc


def Gaussian_psf(taille, w1, w2, phi):
    """utils/Gaussian_psf.m:2-19; [v,u] = ndgrid(x,y): v varies along rows, u along columns."""
    center = (taille + 1) / 2
    xs = [-taille + center + k for k in range(taille)]
    k = M1(taille, taille)
    tot = 0.0
    for i in range(1, taille + 1):
        for j in range(1, taille + 1):
            v, u = xs[i - 1], xs[j - 1]
            U = u * math.cos(phi) - v * math.sin(phi)
            V = u * math.sin(phi) + v * math.cos(phi)
            c = w1 ** 2 * U ** 2 + w2 ** 2 * V ** 2
            val = ((w1 * w2) / (2 * math.pi)) * math.exp(-c / 2)
            k.s(i, j, val)
            tot += val
    for i in range(1, taille + 1):
        for j in range(1, taille + 1):
            k.s(i, j, k.g(i, j) / tot)
    return k


def A_spatial(x, kernel):
    """real(ifft2(fft2(resize(kernel)) .* fft2(x))) written as the circular convolution it is
    (utils/resize.m:8: the taps sit at rows/cols 1..taille of the zero image, i.e. offsets 0..taille-1):
    (A x)(i,j) = sum_{p,q} kernel(p,q) x(i-(p-1), j-(q-1)) with indices mod the image size."""
    m, n, t = x.m, x.n, kernel.m
    out = M1(m, n)
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            acc = 0.0
            for p in range(1, t + 1):
                for q in range(1, t + 1):
                    acc += kernel.g(p, q) * x.g((i - p) % m + 1, (j - q) % n + 1)
            out.s(i, j, acc)
    return out
