// inflate_other.cpp — bzip2 and xz input for the FASTX reader (needletail 0.5 reads both, /root/reference/Cargo.toml:27; call
// sites src/lib.rs:144,221,381, src/cluster.rs:28).  The image ships the run-time libraries (libbz2.so.1.0, liblzma.so.5) but
// not their headers, so the few entry points used are declared here and bound with dlopen at first use; where a library is
// missing the reader refuses the file with a clear message (exit 1: not one of the reference's panics).
#include <dlfcn.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <vector>

#include "../engine.h"

namespace smafa {

namespace {

// ---- libbz2 (bzlib.h, 1.0.x): the streaming decompressor
struct bz_stream_t {
    char *next_in;
    unsigned int avail_in, total_in_lo32, total_in_hi32;
    char *next_out;
    unsigned int avail_out, total_out_lo32, total_out_hi32;
    void *state;
    void *(*bzalloc)(void *, int, int);
    void (*bzfree)(void *, void *);
    void *opaque;
};
constexpr int kBzOk = 0, kBzStreamEnd = 4;
struct BzApi {
    int (*init)(bz_stream_t *, int, int) = nullptr;
    int (*run)(bz_stream_t *) = nullptr;
    int (*end)(bz_stream_t *) = nullptr;
    bool ok = false;
};

// ---- liblzma (lzma/base.h, 5.x): lzma_stream is 136 bytes on LP64; only its first six fields are touched here, the rest
// stays zero (LZMA_STREAM_INIT), in a buffer with room to spare
struct lzma_stream_t {
    const uint8_t *next_in;
    size_t avail_in;
    uint64_t total_in;
    uint8_t *next_out;
    size_t avail_out;
    uint64_t total_out;
    uint64_t rest[26];
};
constexpr int kLzmaRun = 0, kLzmaFinish = 3, kLzmaOk = 0, kLzmaStreamEnd = 1;
constexpr uint32_t kLzmaConcatenated = 0x08;
struct LzmaApi {
    int (*decoder)(lzma_stream_t *, uint64_t, uint32_t) = nullptr;
    int (*code)(lzma_stream_t *, int) = nullptr;
    void (*end)(lzma_stream_t *) = nullptr;
    bool ok = false;
};

template <class F>
bool bind(void *lib, const char *name, F &fn) {
    fn = reinterpret_cast<F>(dlsym(lib, name));
    return fn != nullptr;
}

const BzApi &bz_api() {
    static BzApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = nullptr;
        for (const char *n : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"})
            if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return;
        api.ok = bind(lib, "BZ2_bzDecompressInit", api.init) && bind(lib, "BZ2_bzDecompress", api.run) &&
                 bind(lib, "BZ2_bzDecompressEnd", api.end);
    });
    return api;
}

const LzmaApi &lzma_api() {
    static LzmaApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        void *lib = nullptr;
        for (const char *n : {"liblzma.so.5", "liblzma.so"})
            if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!lib) return;
        api.ok = bind(lib, "lzma_stream_decoder", api.decoder) && bind(lib, "lzma_code", api.code) && bind(lib, "lzma_end", api.end);
    });
    return api;
}

}  // namespace

// the whole decompressed content of a bzip2 file image (concatenated streams included, as bzip2 -d and needletail read them)
int inflate_bzip2(const char *path, const uint8_t *in, size_t n, std::vector<uint8_t> &out) {
    const BzApi &bz = bz_api();
    if (!bz.ok) return set_error(SMAFA_ERR_INVALID, "%s: bzip2 input is not supported by this build (no libbz2 at run time)", path);
    out.clear();
    size_t pos = 0;
    std::vector<uint8_t> chunk(8u << 20);
    while (pos < n) {
        if (pos > 0 && !(n - pos >= 3 && in[pos] == 'B' && in[pos + 1] == 'Z' && in[pos + 2] == 'h')) break;  // trailing bytes
        bz_stream_t s;
        memset(&s, 0, sizeof s);
        if (bz.init(&s, 0, 0) != kBzOk) return set_error(SMAFA_ERR_IO, "%s: bzip2 decoder could not start", path);
        int r = kBzOk;
        while (r == kBzOk) {
            if (s.avail_in == 0 && pos < n) {
                const size_t take = std::min<size_t>(n - pos, 1u << 30);
                s.next_in = (char *)const_cast<uint8_t *>(in + pos);
                s.avail_in = (unsigned)take;
                pos += take;
            }
            s.next_out = (char *)chunk.data();
            s.avail_out = (unsigned)chunk.size();
            const unsigned before_in = s.avail_in;
            r = bz.run(&s);
            const size_t made = chunk.size() - s.avail_out;
            out.insert(out.end(), chunk.begin(), chunk.begin() + made);
            if (r == kBzOk && made == 0 && s.avail_in == before_in && pos >= n) {  // input exhausted inside a stream
                bz.end(&s);
                return set_error(SMAFA_ERR_FORMAT, "%s: truncated bzip2 stream", path);
            }
        }
        pos -= s.avail_in;  // what the finished stream did not consume belongs to the next one
        bz.end(&s);
        if (r != kBzStreamEnd) return set_error(SMAFA_ERR_FORMAT, "%s: damaged bzip2 stream (code %d)", path, r);
    }
    return SMAFA_OK;
}

// the whole decompressed content of an .xz file image (concatenated streams and stream padding included)
int inflate_xz(const char *path, const uint8_t *in, size_t n, std::vector<uint8_t> &out) {
    const LzmaApi &lz = lzma_api();
    if (!lz.ok) return set_error(SMAFA_ERR_INVALID, "%s: xz input is not supported by this build (no liblzma at run time)", path);
    out.clear();
    lzma_stream_t s;
    memset(&s, 0, sizeof s);
    if (lz.decoder(&s, UINT64_MAX, kLzmaConcatenated) != kLzmaOk) return set_error(SMAFA_ERR_IO, "%s: xz decoder could not start", path);
    std::vector<uint8_t> chunk(8u << 20);
    s.next_in = in;
    s.avail_in = n;
    int r = kLzmaOk;
    while (r == kLzmaOk) {
        s.next_out = chunk.data();
        s.avail_out = chunk.size();
        r = lz.code(&s, kLzmaFinish);
        out.insert(out.end(), chunk.begin(), chunk.begin() + (chunk.size() - s.avail_out));
    }
    lz.end(&s);
    if (r != kLzmaStreamEnd) return set_error(SMAFA_ERR_FORMAT, "%s: damaged or truncated xz stream (code %d)", path, r);
    return SMAFA_OK;
}

}  // namespace smafa
