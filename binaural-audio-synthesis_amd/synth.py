"""Bit-reproducible synthetic HRIR / delay-difference tables and trajectories.

The reference's database (`irs_and_delaydiffs_compensated_6.mat`, reference
apply_hrtf.py:23, :602; built by upsample_irs.m:15-54 from the IRCAM LISTEN
set) is not shipped with the reference and cannot be fetched, so every test,
golden vector and benchmark in this repository runs on tables produced here.

Requirements:
  * identical bytes on every machine: only integer draws from
    numpy.random.default_rng plus exact float operations (products and sums of
    small integers and powers of two; IEEE divisions of integers, which are
    correctly rounded everywhere).  No FFT, no libm transcendental.
  * every IR sample is exactly representable in float32, so the float32 device
    copy of the table is lossless.
  * same struct layout as the file the reference loads (apply_hrtf.py:34-44):
    upsampling, diffs_left/right (187,187), irs_left/right (187, 512*U).

Two kinds:
  consistent  - d[i,j] = tau[j]-tau[i] exactly, IR onsets at tau (the "mesh rule"
                of apply_hrtf.py:227-241 holds exactly); used for perf + parity.
  adversarial - random antisymmetric delays including integers and +-40 samples,
                unrelated IRs; used for arithmetic coverage of the shifts.
"""
import hashlib

import numpy as np

N_DIRECTIONS = 187          # upsample_irs.m:16, sphere.py:314
DB_TAPS = 512               # upsample_irs.m:37
DEFAULT_UPSAMPLING = 8      # README.md:46-47

# (elevation deg, number of azimuths) per ring, in database index order
# (sphere.py:127-314 lists the same rings row by row).
RINGS = ((-45, 24), (-30, 24), (-15, 24), (0, 24), (15, 24), (30, 24), (45, 24),
         (60, 12), (75, 6), (90, 1))


def direction_degrees():
    """(187, 2) integer array [elev_deg, azim_deg] in database order."""
    rows = []
    for elev, count in RINGS:
        step = 360 // count
        for i in range(count):
            rows.append((elev, i * step))
    out = np.array(rows, dtype=np.int64)
    assert out.shape == (N_DIRECTIONS, 2)
    return out


def _sin_deg_rational(x):
    """Bhaskara's rational sine approximation for integer degrees.

    Exact-integer numerator/denominator followed by ONE IEEE division, hence
    bit-identical on every machine (no libm).  Accuracy (~1.6e-3) is irrelevant:
    it only shapes the synthetic delays.
    """
    x = np.asarray(x, dtype=np.int64) % 360
    neg = x >= 180
    y = np.where(neg, x - 180, x)
    p = y * (180 - y)
    val = (4 * p).astype(np.float64) / (40500 - p).astype(np.float64)
    return np.where(neg, -val, val)


def _cos_deg_rational(x):
    return _sin_deg_rational(np.asarray(x, dtype=np.int64) + 90)


def _onset_delays_64th():
    """Per-ear onset delay of every direction, in integer 1/64 samples.

    Spherical-head-like: tau = 20 + 14*(1 - cos(angle to the ear axis)) samples,
    the ear axes being azimuth 90 deg (left) / 270 deg (right), elevation 0.
    """
    d = direction_degrees()
    ce, se_az = _cos_deg_rational(d[:, 0]), _sin_deg_rational(d[:, 1])
    c_left = ce * se_az            # cos(angle to left ear axis)
    out = np.empty((2, N_DIRECTIONS), dtype=np.int64)
    for ear, c in enumerate((c_left, -c_left)):
        tau = 20.0 + 14.0 * (1.0 - c)
        out[ear] = np.floor(tau * 64.0 + 0.5).astype(np.int64)
    return out


def _pulse_bank(rng, n_rows, upsampling, onset_up):
    """(n_rows, 512*U) float64 IRs, every value exactly representable in float32.

    A base-rate integer sequence with a power-of-two decaying envelope is
    interpolated to the upsampled rate with an integer triangle kernel
    (piecewise-linear, so the upsampled IR is smooth on the scale of one base
    sample like a really resampled one) and placed at integer upsampled onset.
    """
    u = upsampling
    m = DB_TAPS * u
    n_base = 96
    base = rng.integers(-48, 49, size=(n_rows, n_base)).astype(np.float64)
    base[:, 0] = rng.integers(40, 64, size=n_rows)          # a clear leading peak
    env = 2.0 ** (-(np.arange(n_base) // 12))               # exact powers of two
    base = base * env
    # a weak late tail so that truncation (samples_to_keep) cuts non-zero samples
    tail = rng.integers(-8, 9, size=(n_rows, DB_TAPS)).astype(np.float64) * 2.0 ** -9
    kernel = (u - np.abs(np.arange(-u + 1, u))).astype(np.float64)   # 1..u..1
    out = np.empty((n_rows, m), dtype=np.float64)
    for r in range(n_rows):
        z = np.zeros(m + u * n_base)
        z[int(onset_up[r]) + u * np.arange(n_base)] += base[r]
        z[u * np.arange(DB_TAPS)] += tail[r]
        # every product and partial sum is exact in float64, so the summation
        # order inside np.convolve cannot change the result
        out[r] = np.convolve(z, kernel)[u - 1: u - 1 + m]
    out *= 2.0 ** -9                                        # peak ~ 0.9
    assert np.array_equal(out.astype(np.float32).astype(np.float64), out)
    return out


class SyntheticTable:
    """Plain container with the five fields of the reference's table struct."""

    def __init__(self, upsampling, diffs_left, diffs_right, irs_left, irs_right):
        self.upsampling = int(upsampling)
        self.diffs_left = diffs_left
        self.diffs_right = diffs_right
        self.irs_left = irs_left
        self.irs_right = irs_right

    def truncated(self, samples_to_keep):
        """Same truncation as apply_hrtf.py:43-44."""
        n = samples_to_keep * self.upsampling
        return SyntheticTable(self.upsampling, self.diffs_left, self.diffs_right,
                              self.irs_left[:, :n], self.irs_right[:, :n])

    def sha256(self):
        h = hashlib.sha256()
        h.update(np.int64(self.upsampling).tobytes())
        for a in (self.diffs_left, self.diffs_right, self.irs_left, self.irs_right):
            h.update(np.ascontiguousarray(a, dtype=np.float64).tobytes())
        return h.hexdigest()


def make_table(kind="consistent", seed=0, upsampling=DEFAULT_UPSAMPLING):
    """Full-length (512*U columns) synthetic table.  See module docstring."""
    rng = np.random.default_rng(seed)
    u = int(upsampling)
    if kind == "consistent":
        tau64 = _onset_delays_64th()                        # (2,187) ints
        diffs = [(tau64[e][None, :] - tau64[e][:, None]).astype(np.float64) / 64.0
                 for e in range(2)]
        onsets = [np.floor(tau64[e].astype(np.float64) * u / 64.0 + 0.5).astype(np.int64)
                  for e in range(2)]
    elif kind == "adversarial":
        diffs = []
        for _ in range(2):
            raw = rng.integers(-640, 641, size=(N_DIRECTIONS, N_DIRECTIONS)).astype(np.float64) / 16.0
            raw[rng.integers(0, 4, size=raw.shape) == 0] = np.round(raw[0, 0])   # some exact integers
            up = np.triu(raw, 1)
            up[0, 1], up[1, 2], up[72, 73] = 40.0, -40.0, 3.0
            diffs.append(up - up.T)                          # antisymmetric, zero diagonal
        onsets = [rng.integers(8, 200, size=N_DIRECTIONS) for _ in range(2)]
    else:
        raise ValueError("kind must be 'consistent' or 'adversarial'")
    irs = [_pulse_bank(rng, N_DIRECTIONS, u, onsets[e]) for e in range(2)]
    return SyntheticTable(u, diffs[0], diffs[1], irs[0], irs[1])


def save_table_mat(path, table):
    """Write `table` in the layout load_irs_and_delaydiffs reads (apply_hrtf.py:34-44)."""
    import scipy.io
    scipy.io.savemat(path, {"irs_and_delaydiffs": {
        "upsampling": float(table.upsampling),
        "diffs_left": np.asarray(table.diffs_left, dtype=np.float64),
        "diffs_right": np.asarray(table.diffs_right, dtype=np.float64),
        "irs_left": np.asarray(table.irs_left, dtype=np.float64),
        "irs_right": np.asarray(table.irs_right, dtype=np.float64),
    }}, format="5")


# --------------------------------------------------------------------------
# Trajectory presets of the reference CLI (apply_hrtf.py:580-593), vectorised:
# t may be a scalar or a float64 array of sample times; returns (elev, azim) rad.
# --------------------------------------------------------------------------
def trajectory(name, fs=44100, period_s=4.0, length_s=30.0, turns=15.0, phase=0.0):
    k = 2 * np.pi / (period_s * fs)
    two_pi = 2 * np.pi

    def circle_front(t):
        t = np.asarray(t, dtype=np.float64)
        return np.sin(k * t + phase), np.cos(k * t + phase)

    def circle_horizontal(t):
        t = np.asarray(t, dtype=np.float64)
        return np.zeros_like(t), (k * t + phase) % two_pi

    def circle_askew(t):
        t = np.asarray(t, dtype=np.float64)
        return (np.pi / 4) * np.cos(k * t + phase), (k * t + phase) % two_pi

    def halfcircle_vertical(t):
        t = np.asarray(t, dtype=np.float64)
        c = np.cos(k * t + phase)
        return (np.pi / 2) * (1 - 1.5 * np.abs(c)), (np.pi / 2) * np.sign(c)

    def passing(t):
        t = np.asarray(t, dtype=np.float64)
        return np.zeros_like(t), np.arctan(12 * np.cos(2 * k * t + phase))

    def spiral(t):
        t = np.asarray(t, dtype=np.float64)
        n = fs * length_s
        return (-np.pi / 4) + (3 * np.pi / 4) * (t / n), two_pi * t * turns / n + phase

    table = dict(circle_front=circle_front, circle_horizontal=circle_horizontal,
                 circle_askew=circle_askew, halfcircle_vertical=halfcircle_vertical,
                 passing=passing, spiral=spiral)
    fn = table[name]

    def wrapped(t):
        # scalar t -> np.float64 scalars (the float64 branch of sphere.py:98-119),
        # array t -> float64 arrays
        e, a = fn(t)
        return np.asarray(e, dtype=np.float64)[()], np.asarray(a, dtype=np.float64)[()]

    wrapped.__name__ = name
    return wrapped


def index_function(name, n):
    """Continuous database index over a signal of n samples, for the legacy 1-D path
    (make_signal_move, apply_hrtf.py:294): indices 73..97 are the block its 97 -> 73 wrap serves."""
    table = {
        "ring0_sweep": lambda t: 73.0 + 23.75 * (t / n),              # ends just short of the wrap
        "ring0_wrap": lambda t: 73.0 + (24.0 * 3 * t / n) % 24.0,      # three turns through the wrap
        "low_ring": lambda t: 3.0 + 15.5 * (t / n),                    # inside the lowest ring
    }
    return table[name]


def integer_noise(seed, n, scale=1.0):
    """Seeded white noise in [-1,1) * scale, exactly representable in float32 and
    identical on every machine (integer draws only)."""
    r = np.random.default_rng(seed).integers(-(1 << 15), 1 << 15, size=n)
    return (r.astype(np.float64) * (scale / (1 << 15))).astype(np.float32)
