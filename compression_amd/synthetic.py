"""Seeded synthetic workloads shared by tests/, bench.py and smoke().

Shapes and distributions follow SURVEY.md §8(d) / BASELINE.md §3: discretised
Gaussian tables with sigma_c = 0.25 * 2^(c/24), precision 12, tail_mass 2^-8,
stored in the reference's ragged 1-D layout with NEGATIVE precision (escape
coding enabled, python/entropy_models/continuous_base.py:275-296); symbols are
uniform `precision`-bit draws inverted through each row's CDF.
Pure numpy/scipy — no GPU, no oracle.
"""
from __future__ import annotations

import numpy as np
from scipy.special import ndtr, ndtri


def gaussian_pmfs(num_tables: int = 192, tail_mass: float = 2.0 ** -8, sigma0: float = 0.25,
                  octave: float = 24.0):
    """List of float32 PMF rows (body + overflow mass) and int32 minima (cdf_offset)."""
    pmfs, minima = [], []
    for c in range(num_tables):
        sigma = sigma0 * 2.0 ** (c / octave)
        half = -ndtri(tail_mass / 2) * sigma            # |quantile(tail_mass / 2)|
        lo, hi = int(np.floor(-half)), int(np.ceil(half))
        x = np.arange(lo, hi + 1, dtype=np.float64)
        p = ndtr((x + 0.5) / sigma) - ndtr((x - 0.5) / sigma)
        overflow = max(1.0 - p.sum(), 0.0)
        pmfs.append(np.concatenate([p, [overflow]]).astype(np.float32))
        minima.append(lo)
    return pmfs, np.asarray(minima, np.int32)


def assemble_lookup(cdfs, precision: int, overflow: bool = True) -> np.ndarray:
    """Rows [cdf0=0, ..., 1<<precision] -> ragged 1-D lookup with +-precision headers."""
    head = -precision if overflow else precision
    parts = []
    for c in cdfs:
        parts.append(np.asarray([head], np.int32))
        parts.append(np.asarray(c, np.int32))
    return np.concatenate(parts)


def lookup_rows(lookup: np.ndarray):
    """Splits a VALID ragged 1-D lookup back into (precision, cdf) rows."""
    rows, i = [], 0
    while i < len(lookup):
        sp = int(lookup[i])
        last = 1 << abs(sp)
        j = i + 2
        while lookup[j] != last:
            j += 1
        rows.append((sp, lookup[i + 1:j + 1]))
        i = j + 1
    return rows


def sample_symbols(lookup: np.ndarray, streams: int, elems: int, seed: int = 0,
                   escape_fraction: float = 0.0, escape_seed: int = 1) -> np.ndarray:
    """[streams, elems] int32 symbols, channel j mod ntab, drawn through the row CDFs.
    With escape_fraction > 0 that share of symbols is replaced by out-of-range
    integers +-(len + Geometric(0.2)) (rows with negative precision only)."""
    rows = lookup_rows(lookup)
    ntab = len(rows)
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((streams, elems), np.int32)
    for c, (sp, cdf) in enumerate(rows):
        cols = np.arange(c, elems, ntab)
        if cols.size == 0:
            continue
        u = rng.integers(0, 1 << abs(sp), size=(streams, cols.size), dtype=np.int64)
        sym = np.searchsorted(cdf, u, side="right") - 1
        if sp < 0:
            # the last interval is the escape symbol: fold it onto the previous symbol
            sym = np.minimum(sym, len(cdf) - 3)
            sym = np.maximum(sym, 0)
        out[:, cols] = sym
    if escape_fraction > 0:
        rng2 = np.random.Generator(np.random.PCG64(escape_seed))
        mask = rng2.random((streams, elems)) < escape_fraction
        lens = np.array([len(cdf) - 2 for _, cdf in rows], np.int64)  # escape index per row
        enabled = np.array([sp < 0 for sp, _ in rows])
        col_tab = np.arange(elems) % ntab
        mask &= enabled[col_tab][None, :]
        mag = lens[col_tab][None, :] + rng2.geometric(0.2, size=(streams, elems))
        sign = rng2.integers(0, 2, size=(streams, elems)) * 2 - 1
        esc = np.where(sign > 0, mag, -(mag - lens[col_tab][None, :]))  # >= vmax or < 0
        out = np.where(mask, esc, out).astype(np.int32)
    return out


def empirical_bits(lookup: np.ndarray, symbols: np.ndarray) -> float:
    """Ideal code length in bits of in-range symbols under the tables."""
    rows = lookup_rows(lookup)
    ntab = len(rows)
    bits = 0.0
    for c, (sp, cdf) in enumerate(rows):
        cols = np.arange(c, symbols.shape[1], ntab)
        if cols.size == 0:
            continue
        s = symbols[:, cols].reshape(-1)
        ok = (s >= 0) & (s < len(cdf) - 1)
        q = np.diff(cdf.astype(np.int64))[s[ok]]
        bits += float(np.sum(abs(sp) - np.log2(np.maximum(q, 1))))
    return bits


def lowpass_images(batch: int, height: int, width: int, seed: int = 2) -> np.ndarray:
    """uint8 [batch, height, width, 3] low-pass-filtered noise (stand-in for Kodak-shaped
    images)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    x = rng.standard_normal((batch, height + 16, width + 16, 3)).astype(np.float32)
    k = np.hanning(17).astype(np.float32)
    k /= k.sum()
    for axis in (1, 2):
        x = np.apply_along_axis(lambda v: np.convolve(v, k, mode="valid"), axis, x)
    x = (x - x.min()) / (x.max() - x.min())
    return np.round(255 * x).astype(np.uint8)
