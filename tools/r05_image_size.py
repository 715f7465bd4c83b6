import sys; sys.path.insert(0, '.')
import numpy as np, torch
import compression_amd as tfc
from compression_amd import synthetic
torch.manual_seed(0)
for name, make in (("bls2017", lambda: tfc.models.BLS2017Model(num_filters=192).cuda()), ("bmshj2018", lambda: tfc.models.BMSHJ2018Model(num_filters=192).cuda())):
    m = make().init_compression()
    for em in [e for e in (m.entropy_model, getattr(m, "side_entropy_model", None)) if e is not None]:
        rows = synthetic.lookup_rows(em.cdf.cpu().numpy())
        nsyms = np.array([len(c) - 1 for _, c in rows])
        prec = abs(rows[0][0])
        ntab = len(rows)
        cdf_entries = int((nsyms + 1).sum()) + 3
        words = (ntab + 1) * max(1, (1 << prec) // 64)
        dirb = 16 * (ntab + 17)
        cdfb = (2 * cdf_entries + 15) & ~15
        enc = dirb + cdfb
        dec = enc + 8 * words + ((2 * words + 15) & ~15)
        print(name, type(em).__name__, "ntab", ntab, "prec", prec, "nsym max/mean/sum", nsyms.max(), round(nsyms.mean(), 1), nsyms.sum(),
              "dir", dirb, "cdf", cdfb, "bitmaps", 8 * words, "counts", 2 * words, "image", dec, "u8 counts:", dec - words)
