"""CPU restatement of the reference's randomised-fern keyframe database (TEST INFRASTRUCTURE — only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

Follows elasticfusion/Core/src/Ferns.cpp line by line: generateFerns (:66-84), addFrame (:170-276) with its inverted
co-occurrence lists, findFrame (:277-423), photometricCheck (:604-668), blockHDAware (:684-704), consume (:160-168);
Resize::image / vertex (Shaders/Resize.cpp:67-143, resize.frag) as NEAREST sampling at pixel centres.

PARITY UNPINNED: the reference has no test or fixture for this class and seeds its table with time(0) (Ferns.cpp:56);
the table here is mt19937(seed) through Lemire's nearly-divisionless uniform mapping (libstdc++ 11
uniform_int_distribution over a 32-bit generator), the mt19937 itself is pinned by the standard's known answer
(10000th output of seed 5489 = 4123659995, [rand.predef]).
"""
import numpy as np

BAD = 255


class MT19937:
    def __init__(self, seed):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def next(self):
        if self.idx >= 624:
            mt = self.mt
            for i in range(624):
                y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF

    def uniform(self, a, b):
        rng = (b - a) + 1
        product = self.next() * rng
        low = product & 0xFFFFFFFF
        if low < rng:
            threshold = ((1 << 32) - rng) % rng
            while low < threshold:
                product = self.next() * rng
                low = product & 0xFFFFFFFF
        return a + (product >> 32)


def _texel(u, n):
    """NEAREST / CLAMP_TO_EDGE texel of a normalised coordinate, the product evaluated in float32."""
    t = int(np.floor(np.float32(u) * np.float32(n)))
    return min(max(t, 0), n - 1)


def resize_nearest(src, th, tw):
    rows, cols = src.shape[:2]
    ys = [_texel((np.float32(j) + np.float32(0.5)) / np.float32(th), rows) for j in range(th)]
    xs = [_texel((np.float32(i) + np.float32(0.5)) / np.float32(tw), cols) for i in range(tw)]
    return np.ascontiguousarray(src[np.ix_(ys, xs)])


class Frame:
    def __init__(self, fid, pose, srcTime, codes, good, rgb, verts, norms):
        self.id, self.pose, self.srcTime = fid, np.array(pose, np.float32).reshape(4, 4), srcTime
        self.codes, self.goodCodes = codes, good
        self.initRgb, self.initVerts, self.initNorms = rgb, verts, norms


class Ferns:
    def __init__(self, width, height, K, num=500, maxDepth_mm=3000, photoThresh=115.0, seed=0, make_odometry=None):
        self.num, self.factor = num, 8
        self.W, self.H = width, height
        self.width, self.height = width // 8, height // 8
        self.maxDepth, self.photoThresh = maxDepth_mm, np.float32(photoThresh)
        self.fx, self.fy, self.cx, self.cy = [np.float32(v) for v in K]
        rng = MT19937(seed)
        self.pos = np.zeros((num, 2), np.int32)
        self.rgbd = np.zeros((num, 4), np.int32)
        for i in range(num):  # Ferns.cpp:68-83
            self.pos[i, 0] = rng.uniform(0, self.width - 1)
            self.pos[i, 1] = rng.uniform(0, self.height - 1)
            self.rgbd[i, 0] = rng.uniform(0, 255)
            self.rgbd[i, 1] = rng.uniform(0, 255)
            self.rgbd[i, 2] = rng.uniform(0, 255)
            self.rgbd[i, 3] = rng.uniform(400, maxDepth_mm)
        self.ids = [[[] for _ in range(16)] for _ in range(num)]  # conservatory[i].ids[code]
        self.frames = []
        self.lastClosest = -1
        self._make_odometry = make_odometry  # () -> oracle RGBDOdometry at thumbnail size

    # -- encoding of one frame's thumbnails -----------------------------------------------------
    def _encode(self, img, verts):
        codes = np.full(self.num, BAD, np.uint8)
        good = 0
        co = np.zeros(len(self.frames), np.int64)
        for i in range(self.num):
            x, y = int(self.pos[i, 0]), int(self.pos[i, 1])
            z = verts[y, x, 2]
            if z > 0:
                pix = img[y, x]
                code = ((int(pix[0]) > self.rgbd[i, 0]) << 3) | ((int(pix[1]) > self.rgbd[i, 1]) << 2) | ((int(pix[2]) > self.rgbd[i, 2]) << 1) | \
                       int(int(np.float32(z) * np.float32(1000.0)) > self.rgbd[i, 3])
                good += 1
                for j in self.ids[i][code]:
                    co[j] += 1
                codes[i] = code
        return codes, good, co

    def _thumbs(self, image, vertex, normal):
        return (resize_nearest(image, self.height, self.width), resize_nearest(vertex, self.height, self.width),
                resize_nearest(normal, self.height, self.width))

    def _add(self, img, verts, norms, pose, srcTime, threshold):
        codes, good, co = self._encode(img, verts)
        minimum = np.float32(3.402823466e+38)
        if good > 0:
            for i, fr in enumerate(self.frames):
                maxCo = np.float32(min(good, fr.goodCodes))
                dissim = np.float32(maxCo - np.float32(co[i])) / maxCo
                if dissim < minimum:
                    minimum = dissim
        if (minimum > np.float32(threshold) or len(self.frames) == 0) and good > 0:
            fid = len(self.frames)
            for i in range(self.num):
                if codes[i] != BAD:
                    self.ids[i][codes[i]].append(fid)
            self.frames.append(Frame(fid, pose, srcTime, codes, good, img.copy(), verts.copy(), norms.copy()))
            return True
        return False

    def addFrame(self, image, vertex, normal, pose, srcTime, threshold):
        img, verts, norms = self._thumbs(image, vertex, normal)
        return self._add(img, verts, norms, pose, srcTime, threshold)

    def consume(self, other, relativeTransform, threshold):
        T = np.asarray(relativeTransform, np.float32).reshape(4, 4)
        added = 0
        for fr in other.frames:
            added += int(self._add(fr.initRgb, fr.initVerts, fr.initNorms, _mul44(T, fr.pose), fr.srcTime, threshold))
        return added

    @staticmethod
    def blockHDAware(c1, c2):
        both = (c1 != BAD) & (c2 != BAD)
        return np.float32((c1[both] == c2[both]).sum()) / np.float32(both.sum())

    def photometricCheck(self, vertSmall, imgSmall, estPose, fernPose, fernRgb):
        cx, cy = self.cx / np.float32(8), self.cy / np.float32(8)
        invfx = np.float32(1.0) / np.float32(self.fx / np.float32(8))
        invfy = np.float32(1.0) / np.float32(self.fy / np.float32(8))
        diff = _mul44(_inv44(fernPose), estPose)
        photoSum, photoCount = np.float32(0), 0
        for i in range(self.num):
            x, y = int(self.pos[i, 0]), int(self.pos[i, 1])
            v = vertSmall[y, x]
            if v[2] > 0 and int(np.float32(v[2]) * np.float32(1000.0)) < self.maxDepth:
                w = _mul4v(diff, v)
                with np.errstate(all="ignore"):
                    c0f = w[0] * (np.float32(1) / invfx) / w[2] + cx
                    c1f = w[1] * (np.float32(1) / invfy) / w[2] + cy
                if not (np.isfinite(c0f) and np.isfinite(c1f)):
                    continue
                c0, c1 = int(c0f), int(c1f)
                if 0 <= c0 < self.width and 0 <= c1 < self.height and fernRgb[c1, c0, :3].max() > 0:
                    for ch in range(3):
                        photoSum = np.float32(photoSum + np.float32(abs(int(fernRgb[c1, c0, ch]) - int(imgSmall[y, x, ch]))))
                    photoCount += 1
        with np.errstate(all="ignore"):
            return np.float32(photoSum) / np.float32(photoCount)

    def _search(self, img, verts, time, interMap):
        """Ferns.cpp:327-339: the stored frame of minimum dissimilarity (-1: none eligible)"""
        codes, good, co = self._encode(img, verts)
        minimum, minId = np.float32(3.402823466e+38), -1
        for i, fr in enumerate(self.frames):
            maxCo = np.float32(min(good, fr.goodCodes))
            with np.errstate(all="ignore"):
                dissim = np.float32(maxCo - np.float32(co[i])) / maxCo
            if dissim < minimum and (interMap or (time - fr.srcTime > 300)):
                minimum, minId = dissim, i
        return codes, minimum, minId

    def searchHit(self, thumbs, time, interMap=True):
        """The first half of findFrame, up to the test that lets the tracker verify at all (Ferns.cpp:327-342): True when a candidate
        exists and blockHDAware > 0.3 against it.  (What the pipelined session evaluates every tick, dms_ferns_search_blocks_hd.)"""
        img, verts, _ = thumbs
        codes, _, minId = self._search(img, verts, time, interMap)
        if minId == -1:
            return False
        with np.errstate(all="ignore"):
            return bool(float(self.blockHDAware(codes, self.frames[minId].codes)) > 0.3)  # float against the DOUBLE literal (Ferns.cpp:346)

    def findFrame(self, currPose, vertex, normal, image, time, lost=False, interMap=False, thumbs=None):
        """returns dict(closest, candidate, dissimilarity, blockHDAware, icp_error, icp_count, photo_error, estPose, constraints)"""
        self.lastClosest = -1
        img, verts, norms = thumbs if thumbs is not None else self._thumbs(image, vertex, normal)
        codes, minimum, minId = self._search(img, verts, time, interMap)
        out = dict(closest=-1, candidate=minId, dissimilarity=float(minimum), blockHDAware=0.0, icp_error=0.0, icp_count=0.0, photo_error=0.0,
                   estPose=np.eye(4, dtype=np.float32), constraints=np.zeros((0, 8), np.float32))
        if minId == -1:
            return out
        hd = self.blockHDAware(codes, self.frames[minId].codes)
        out["blockHDAware"] = float(hd)
        if not float(hd) > 0.3:  # Ferns.cpp:346: the float return value against the double literal 0.3 (0.3f itself passes)
            return out
        fr = self.frames[minId]
        fernPose = fr.pose
        cutoff = np.float32(self.maxDepth) / np.float32(1000.0)
        o = self._make_odometry()
        o.initICPModel(fr.initVerts, fr.initNorms, float(cutoff), fernPose)
        o.initICPMaps(verts, norms, float(cutoff))
        deep = int(interMap) == 1  # interMap == 2: candidates of any age, verified with the intra-map tracker settings (dmslam_ferns.h)
        t, R, res = o.getIncrementalTransformation(fernPose[:3, 3].copy(), fernPose[:3, :3].copy(), rgbOnly=False, icpWeight=100.0,
                                                   pyramid=deep, fastOdom=False, so3=deep, interMap=deep)
        est = np.eye(4, dtype=np.float32)
        est[:3, :3], est[:3, 3] = R, t
        out["estPose"] = est
        out["icp_error"], out["icp_count"] = float(res.lastICPError), float(res.lastICPCount)
        photo = self.photometricCheck(verts, img, est, fernPose, fr.initRgb)
        out["photo_error"] = float(photo)
        if res.lastICPError < np.float32(0.0003) and res.lastICPCount > 400 and photo < self.photoThresh:
            self.lastClosest = out["closest"] = minId
            cons = []
            step = max(1, self.num // 50)
            cp = np.asarray(currPose, np.float32).reshape(4, 4)
            for i in range(0, self.num, step):
                x, y = int(self.pos[i, 0]), int(self.pos[i, 1])
                v = verts[y, x]
                if v[2] > 0 and int(np.float32(v[2]) * np.float32(1000.0)) < self.maxDepth:
                    cons.append(np.concatenate([_mul4v(cp, v), _mul4v(est, v)]))
            out["constraints"] = np.array(cons, np.float32).reshape(-1, 8)
        return out


def _mul44(a, b):
    a, b = np.asarray(a, np.float32).reshape(4, 4), np.asarray(b, np.float32).reshape(4, 4)
    o = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            s = np.float32(a[i, 0] * b[0, j])
            s = np.float32(s + np.float32(a[i, 1] * b[1, j]))
            s = np.float32(s + np.float32(a[i, 2] * b[2, j]))
            s = np.float32(s + np.float32(a[i, 3] * b[3, j]))
            o[i, j] = s
    return o


def _mul4v(m, v):
    m = np.asarray(m, np.float32).reshape(4, 4)
    o = np.zeros(4, np.float32)
    for i in range(4):
        s = np.float32(m[i, 0] * v[0])
        s = np.float32(s + np.float32(m[i, 1] * v[1]))
        s = np.float32(s + np.float32(m[i, 2] * v[2]))
        s = np.float32(s + m[i, 3])
        o[i] = s
    return o


def _inv44(m):
    """general 4x4 inverse by cofactors in float32, one reciprocal of the determinant (the product's sm::inv4t<float>)"""
    m = np.asarray(m, np.float32).reshape(16)
    f = np.float32
    inv = np.zeros(16, np.float32)

    def t3(a, b, c):
        return f(f(m[a] * m[b]) * m[c])

    def six(p):
        s = f(0)
        first = True
        for sign, (a, b, c) in p:
            term = t3(a, b, c)
            if first:
                s = term if sign > 0 else f(-term)
                first = False
            else:
                s = f(s + term) if sign > 0 else f(s - term)
        return s

    # cofactor expansion, term order of sm::inv4t (smallmath.hpp)
    inv[0] = six([(1, (5, 10, 15)), (-1, (5, 11, 14)), (-1, (9, 6, 15)), (1, (9, 7, 14)), (1, (13, 6, 11)), (-1, (13, 7, 10))])
    inv[4] = six([(-1, (4, 10, 15)), (1, (4, 11, 14)), (1, (8, 6, 15)), (-1, (8, 7, 14)), (-1, (12, 6, 11)), (1, (12, 7, 10))])
    inv[8] = six([(1, (4, 9, 15)), (-1, (4, 11, 13)), (-1, (8, 5, 15)), (1, (8, 7, 13)), (1, (12, 5, 11)), (-1, (12, 7, 9))])
    inv[12] = six([(-1, (4, 9, 14)), (1, (4, 10, 13)), (1, (8, 5, 14)), (-1, (8, 6, 13)), (-1, (12, 5, 10)), (1, (12, 6, 9))])
    inv[1] = six([(-1, (1, 10, 15)), (1, (1, 11, 14)), (1, (9, 2, 15)), (-1, (9, 3, 14)), (-1, (13, 2, 11)), (1, (13, 3, 10))])
    inv[5] = six([(1, (0, 10, 15)), (-1, (0, 11, 14)), (-1, (8, 2, 15)), (1, (8, 3, 14)), (1, (12, 2, 11)), (-1, (12, 3, 10))])
    inv[9] = six([(-1, (0, 9, 15)), (1, (0, 11, 13)), (1, (8, 1, 15)), (-1, (8, 3, 13)), (-1, (12, 1, 11)), (1, (12, 3, 9))])
    inv[13] = six([(1, (0, 9, 14)), (-1, (0, 10, 13)), (-1, (8, 1, 14)), (1, (8, 2, 13)), (1, (12, 1, 10)), (-1, (12, 2, 9))])
    inv[2] = six([(1, (1, 6, 15)), (-1, (1, 7, 14)), (-1, (5, 2, 15)), (1, (5, 3, 14)), (1, (13, 2, 7)), (-1, (13, 3, 6))])
    inv[6] = six([(-1, (0, 6, 15)), (1, (0, 7, 14)), (1, (4, 2, 15)), (-1, (4, 3, 14)), (-1, (12, 2, 7)), (1, (12, 3, 6))])
    inv[10] = six([(1, (0, 5, 15)), (-1, (0, 7, 13)), (-1, (4, 1, 15)), (1, (4, 3, 13)), (1, (12, 1, 7)), (-1, (12, 3, 5))])
    inv[14] = six([(-1, (0, 5, 14)), (1, (0, 6, 13)), (1, (4, 1, 14)), (-1, (4, 2, 13)), (-1, (12, 1, 6)), (1, (12, 2, 5))])
    inv[3] = six([(-1, (1, 6, 11)), (1, (1, 7, 10)), (1, (5, 2, 11)), (-1, (5, 3, 10)), (-1, (9, 2, 7)), (1, (9, 3, 6))])
    inv[7] = six([(1, (0, 6, 11)), (-1, (0, 7, 10)), (-1, (4, 2, 11)), (1, (4, 3, 10)), (1, (8, 2, 7)), (-1, (8, 3, 6))])
    inv[11] = six([(-1, (0, 5, 11)), (1, (0, 7, 9)), (1, (4, 1, 11)), (-1, (4, 3, 9)), (-1, (8, 1, 7)), (1, (8, 3, 5))])
    inv[15] = six([(1, (0, 5, 10)), (-1, (0, 6, 9)), (-1, (4, 1, 10)), (1, (4, 2, 9)), (1, (8, 1, 6)), (-1, (8, 2, 5))])
    det = f(f(f(f(m[0] * inv[0]) + f(m[1] * inv[4])) + f(m[2] * inv[8])) + f(m[3] * inv[12]))
    idet = f(f(1) / det)
    return (inv * idet).astype(np.float32).reshape(4, 4)
