// Host-side segment reader for the on-disk format of the reference's pre-generated data sets:
// `{idx}.wav` = RIFF/WAVE, 16 kHz, nch channels, 16-bit PCM (written by soundfile.write, code/data_generation/
// utils_simu_rir_sig.py:855-856; read back per item by soundfile.read in FixMicSigDataset.__getitem__, code/dataset.py:147-151).
//
// A training step at 8 GPUs consumes 512 files x 263 KB; Python worker processes spend most of their time in pickling and
// per-file float conversion.  Here one call fills a whole (pinned) batch buffer with raw int16 PCM using a few POSIX threads
// doing positional reads; the int16 -> f32 conversion happens inside the STFT kernel on the device.  No GPU calls in this file.
#include "common.h"
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct WavInfo { int nch, fs, bits, tag; long data_off, data_bytes; };

static uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// walk the RIFF chunks with positional reads; returns 0 or a negative code (message in err)
static int wav_parse(int fd, WavInfo* w, char* err, size_t errn, const char* path) {
    unsigned char h[12];
    if (pread(fd, h, 12, 0) != 12 || memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) {
        snprintf(err, errn, "%s: not a RIFF/WAVE file", path); return -1;
    }
    struct stat stt; fstat(fd, &stt);
    long pos = 12; bool have_fmt = false, have_data = false;
    while (pos + 8 <= stt.st_size && !(have_fmt && have_data)) {
        unsigned char c[8];
        if (pread(fd, c, 8, pos) != 8) break;
        const long size = rd32(c + 4);
        if (!memcmp(c, "fmt ", 4)) {
            unsigned char f[40]; const long n = size < 40 ? size : 40;
            if (n < 16 || pread(fd, f, n, pos + 8) != n) { snprintf(err, errn, "%s: truncated fmt chunk", path); return -1; }
            w->tag = rd16(f); w->nch = rd16(f + 2); w->fs = (int)rd32(f + 4); w->bits = rd16(f + 14);
            if (w->tag == 0xFFFE && n >= 26) w->tag = rd16(f + 24);              // WAVE_FORMAT_EXTENSIBLE: sub-format tag
            have_fmt = true;
        } else if (!memcmp(c, "data", 4)) {
            w->data_off = pos + 8;
            w->data_bytes = size;
            if (w->data_off + w->data_bytes > stt.st_size) w->data_bytes = stt.st_size - w->data_off;   // streamed writers leave 0 / -1
            have_data = true;
        }
        pos += 8 + size + (size & 1);
    }
    if (!have_fmt || !have_data) { snprintf(err, errn, "%s: missing fmt/data chunk", path); return -1; }
    if (w->tag != 1 || w->bits != 16) { snprintf(err, errn, "%s: only 16-bit PCM is supported (format tag %d, %d bits)", path, w->tag, w->bits); return -1; }
    return 0;
}

extern "C" int sarssl_wav_probe(const char* path, int* nch, int* fs, long* nsample) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { sarssl_set_error("%s: cannot open", path); return -1; }
    WavInfo w; char err[512];
    const int rc = wav_parse(fd, &w, err, sizeof(err), path);
    close(fd);
    if (rc) { sarssl_set_error("%s", err); return rc; }
    *nch = w.nch; *fs = w.fs; *nsample = w.data_bytes / (2L * w.nch);
    return 0;
}

// out[i][0..nsample)[0..nch) = samples [offset, offset + nsample) of paths[i], raw int16.  Every file must have nch channels,
// sample rate fs (0 = do not check) and at least offset + nsample samples.  nthreads <= 0 -> min(n, 8).
extern "C" int sarssl_wav_read_batch(const char* const* paths, int n, long nsample, int nch, int fs, long offset, int16_t* out,
                                     int nthreads) {
    SARSSL_REQUIRE(n >= 0 && nsample > 0 && nch > 0 && offset >= 0 && out, "sarssl_wav_read_batch");
    if (n == 0) return 0;
    if (nthreads <= 0) nthreads = n < 8 ? n : 8;
    if (nthreads > n) nthreads = n;
    std::atomic<int> next(0), failed(0);
    std::vector<std::string> errs(nthreads);
    const size_t bytes = (size_t)nsample * nch * 2;
    auto work = [&](int tid) {
        char err[512];
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= n || failed.load()) return;
            const int fd = open(paths[i], O_RDONLY);
            if (fd < 0) { errs[tid] = std::string(paths[i]) + ": cannot open"; failed = 1; return; }
            WavInfo w;
            int rc = wav_parse(fd, &w, err, sizeof(err), paths[i]);
            if (!rc && w.nch != nch) { snprintf(err, sizeof(err), "%s: %d channels, expected %d", paths[i], w.nch, nch); rc = -1; }
            if (!rc && fs > 0 && w.fs != fs) { snprintf(err, sizeof(err), "%s: sample rate %d, expected %d", paths[i], w.fs, fs); rc = -1; }
            if (!rc && (offset + nsample) * 2L * nch > w.data_bytes) {
                snprintf(err, sizeof(err), "%s: %ld samples, need %ld", paths[i], w.data_bytes / (2L * nch), offset + nsample); rc = -1;
            }
            if (!rc) {
                char* dst = (char*)out + (size_t)i * bytes;
                size_t got = 0;
                while (got < bytes) {
                    const ssize_t r = pread(fd, dst + got, bytes - got, w.data_off + offset * 2L * nch + (long)got);
                    if (r <= 0) { snprintf(err, sizeof(err), "%s: short read", paths[i]); rc = -1; break; }
                    got += (size_t)r;
                }
            }
            close(fd);
            if (rc) { errs[tid] = err; failed = 1; return; }
        }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    if (failed.load()) {
        for (auto& e : errs) if (!e.empty()) { sarssl_set_error("%s", e.c_str()); break; }
        return -1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ mask sampling (host)
// PatchMask.forward (code/common/utils_module.py:263-267, 305-308) draws, per batch item, random.sample(range(npatch), nmasked)
// then random.randint(0, nmic-1) from Python's global Mersenne Twister.  64 x sample(256, 128) costs 3.5 ms of interpreter time
// per step; this restates exactly what CPython does with the generator state (random.getstate() -> 624 words + position):
//   genrand_uint32 (MT19937), getrandbits(k <= 32) = genrand >> (32 - k), _randbelow(n) = rejection on n.bit_length() bits,
//   sample(): pool-based selection (n <= setsize: always true for npatch <= 1045 when nmasked > 5), randint(a,b) = a + _randbelow(b-a+1)
// so the stream of masks is bit-identical to the reference's for the same seed.  mt: 624 state words, *pos in/out.
static inline uint32_t mt_genrand(uint32_t* mt, int* pos) {
    if (*pos >= 624) {
        static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
        int kk; uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) { y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 1u]; }
        for (; kk < 623; kk++) { y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 1u]; }
        y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu); mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 1u];
        *pos = 0;
    }
    uint32_t y = mt[(*pos)++];
    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
    return y;
}
static inline int bit_length(uint32_t n) { int k = 0; while (n) { ++k; n >>= 1; } return k; }
static inline uint32_t mt_randbelow(uint32_t* mt, int* pos, uint32_t n) {
    const int k = bit_length(n);
    uint32_t r = mt_genrand(mt, pos) >> (32 - k);
    while (r >= n) r = mt_genrand(mt, pos) >> (32 - k);
    return r;
}
extern "C" int sarssl_mask_sample(unsigned int* mt, int* pos, int nbatch, int npatch, int nmasked, int nmic, long* idx, long* ch) {
    SARSSL_REQUIRE(mt && pos && idx && ch && nbatch >= 0 && npatch > 0 && npatch <= 1045 && nmasked > 5 && nmasked <= npatch && nmic >= 1,
                   "sarssl_mask_sample(5 < nmasked <= npatch <= 1045)");
    std::vector<long> pool(npatch);
    for (int b = 0; b < nbatch; ++b) {
        for (int i = 0; i < npatch; ++i) pool[i] = i;
        for (int i = 0; i < nmasked; ++i) {
            const uint32_t j = mt_randbelow(mt, pos, (uint32_t)(npatch - i));
            idx[(long)b * nmasked + i] = pool[j];
            pool[j] = pool[npatch - i - 1];
        }
        ch[b] = (long)mt_randbelow(mt, pos, (uint32_t)nmic);          // randint(0, nmic-1) = randrange(0, nmic)
    }
    return 0;
}
