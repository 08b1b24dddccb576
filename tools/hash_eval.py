import numpy as np
M32=np.uint64(0xFFFFFFFF)
def mix32(x):
    x=x.astype(np.uint64)
    x^=x>>np.uint64(16); x=(x*np.uint64(0x85EBCA6B))&M32
    x^=x>>np.uint64(13); x=(x*np.uint64(0xC2B2AE35))&M32
    x^=x>>np.uint64(16); return x
def mul24(a,c):  # low 24 bits of a times 24-bit constant, low 32 bits of the product
    return ((a&np.uint64(0xFFFFFF))*np.uint64(c))&M32
def fast(x, C1=0xE35A2B, C2=0xB5297B, s1=16, s2=15, s3=16):
    x=x.astype(np.uint64)
    x^=x>>np.uint64(s1); x=mul24(x,C1)
    x^=x>>np.uint64(s2); x=mul24(x,C2)
    x^=x>>np.uint64(s3); return x
def stats(h, name, key):
    n=2_000_000
    idx=np.arange(n,dtype=np.uint64)
    v=h(idx^np.uint64(key))
    lo=(v&np.uint64(0xFFFF)).astype(np.int64); hi=(v>>np.uint64(16)).astype(np.int64)
    t=int(0.2*65536)
    keep_lo=(lo>=t); keep_hi=(hi>=t)
    # keep fraction, correlation lo/hi, adjacent correlation, bit balance, chi2 over 256 buckets of top byte and low byte
    def corr(a,b): return float(np.corrcoef(a.astype(float),b.astype(float))[0,1])
    bits=[( (v>>np.uint64(b))&np.uint64(1)).mean() for b in range(32)]
    chi=lambda x: float(((np.bincount(x,minlength=256)-n/256)**2/(n/256)).sum())
    print(name, 'key %08x'%key, 'keep lo %.5f hi %.5f'%(keep_lo.mean(),keep_hi.mean()), 'corr lo/hi %.4f'%corr(keep_lo,keep_hi), 'adj lo %.4f hi %.4f'%(corr(keep_lo[:-1],keep_lo[1:]),corr(keep_hi[:-1],keep_hi[1:])),
          'adj64 %.4f'%corr(keep_lo[:-64],keep_lo[64:]), 'bit dev max %.4f'%max(abs(b-0.5) for b in bits), 'chi2 lo8 %.0f hi8 %.0f (256 dof)'%(chi((lo&255)),chi((hi>>8))))
    # 2-D structure: rows of 64 columns, keep-rate per column / per row variance vs binomial
    k=keep_lo[:64*30000].reshape(-1,64)
    print('   col keep std %.5f (binomial %.5f)  row keep std %.5f (binomial %.5f)'%(k.mean(0).std(), np.sqrt(0.2*0.8/30000), k.mean(1).std(), np.sqrt(0.2*0.8/64)))
for key in (0x1234567, 0xDEADBEEF, 0x0, 0x9E3779B9):
    stats(mix32,'mix32',key); stats(fast,'fast24',key)
print('---- mad24 variant')
def fast2(x, C1=0xE35A2B, C2=0xB5297B):
    x=x.astype(np.uint64)
    h=x^(x>>np.uint64(16)); h=(mul24(h,C1)+x)&M32
    h^=h>>np.uint64(15); h=mul24(h,C2)
    h^=h>>np.uint64(16); return h
for key in (0x1234567, 0xDEADBEEF, 0x0):
    stats(fast2,'mad24',key)
# collision structure of the plain variant vs the mad variant
rng=np.random.default_rng(1)
x=rng.integers(0,2**32,size=200000,dtype=np.uint64)
for b in (1,37,255):
    d=np.uint64((b<<24)|(b<<8))
    print('b',b,'fast24 equal frac',float((fast(x)==fast(x^d)).mean()),'mad24 equal frac',float((fast2(x)==fast2(x^d)).mean()))
# avalanche: flip each input bit, fraction of output bits flipped
for name,h in (('mix32',mix32),('mad24',fast2)):
    av=[]
    for bit in range(32):
        y=h(x)^h(x^np.uint64(1<<bit))
        av.append(np.unpackbits(y.astype('>u4').view(np.uint8)).mean()*1.0)
    print(name,'avalanche min %.3f max %.3f'%(min(av),max(av)))
# sequential-index 16-bit halves: birthday-ish duplicate count
v=fast2(np.arange(4_000_000,dtype=np.uint64)^np.uint64(0x1234567)); print('distinct outputs of 4M sequential inputs:',len(np.unique(v)))
v=mix32(np.arange(4_000_000,dtype=np.uint64)^np.uint64(0x1234567)); print('mix32:',len(np.unique(v)))


def quad_stream_report():
    """Round 5: MaskEval::elem_mult_quad (gemm.hpp) -- four keep / drop decisions from one focal_hash24 word and a three-instruction
    second word.  Rates, pairwise correlations of the four decisions and lag-1 / lag-256 correlations over 4 M quads at p = 0.1 / 0.2 / 0.5."""
    import numpy as np

    def hash24(x):
        x = x.astype(np.uint32)
        h = x ^ (x >> np.uint32(16))
        h = ((h & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0xE35A2B)).astype(np.uint32) + x
        h ^= h >> np.uint32(15)
        h = ((h & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0xB5297B)).astype(np.uint32)
        return h ^ (h >> np.uint32(16))

    def second(h):
        g = h ^ (h >> np.uint32(13))
        g = ((g & np.uint32(0xFFFFFF)).astype(np.uint64) * np.uint64(0xC2B2AF)).astype(np.uint32) + np.uint32(0x165667B1)
        return g ^ (g >> np.uint32(15))
    idx = np.arange(1 << 22, dtype=np.uint32)
    for p in (0.1, 0.2, 0.5):
        t16 = np.uint32(int(p * 16777216) >> 8)
        h = hash24(idx ^ np.uint32(0x9E3779B9))
        g = second(h)
        d = np.stack([(h & 0xffff) < t16, (h >> 16) < t16, (g & 0xffff) < t16, (g >> 16) < t16]).astype(np.float64)
        print(f"quad stream p = {p}: rates {d.mean(1).round(4)}, pairwise corr {np.corrcoef(d)[np.triu_indices(4, 1)].round(4)}")
        for lag in (1, 256):
            print(f"   lag {lag}: {[round(float(np.corrcoef(d[i][:-lag], d[i][lag:])[0, 1]), 4) for i in range(4)]}")


if __name__ == "__main__" and "--quad" in __import__("sys").argv:
    quad_stream_report()
