// Error channel, ABI version and host-side checksum of libmonopsr_hip.so.
#include "common.h"

namespace mpsr {
char *error_buffer()
{
    static thread_local char buf[kErrorBufferBytes] = {0};
    return buf;
}
thread_local int t_call_math = -1;
thread_local int t_call_wino_policy = -1;
}  // namespace mpsr

extern "C" const char *mpsr_last_error(void) { return mpsr::error_buffer(); }
extern "C" int mpsr_abi_version(void) { return 6; }

// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8 on the host.  Used by the TensorFlow
// checkpoint reader/writer (core/tf_checkpoint.py) to verify block trailers and tensor payloads.
namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables()
    {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
};
}  // namespace

extern "C" uint32_t mpsr_crc32c(uint32_t crc, const void *data, size_t n)
{
    static const Crc32cTables T;
    const unsigned char *p = static_cast<const unsigned char *>(data);
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        __builtin_memcpy(&lo, p, 4);
        __builtin_memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T.t[7][lo & 0xff] ^ T.t[6][(lo >> 8) & 0xff] ^ T.t[5][(lo >> 16) & 0xff] ^ T.t[4][lo >> 24] ^
            T.t[3][hi & 0xff] ^ T.t[2][(hi >> 8) & 0xff] ^ T.t[1][(hi >> 16) & 0xff] ^ T.t[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xff];
    return ~c;
}
