// alphabet.cpp — byte -> symbol code tables.
//
// Nucleotides follow the classes of the reference's BYTE_LUT (/root/reference/src/lib.rs:171-178):
// A/a, C/c, G/g, T/t/U/u are four symbols; N, the IUPAC ambiguity letters W S M K R Y B D H V and '-'
// (either case) collapse into ONE fifth symbol that matches itself; every other byte is rejected
// (src/lib.rs:35-42).  Codes here are 0..4 instead of a 5-bit one-hot: the device compares codes by
// bit-planes, and any injective code gives the same mismatch count.
// The amino-acid alphabet is this build's extension (the reference cannot take amino acids): letters
// fold to upper case, every letter plus '*' and '-' is an ordinary symbol that matches only itself.
#include "../engine.h"

namespace smafa {

namespace {
struct Tables {
    uint8_t nt[256];
    uint8_t aa[256];
    Tables() {
        for (int i = 0; i < 256; i++) nt[i] = aa[i] = 255;
        const char *classes[5] = {"Aa", "Cc", "Gg", "TtUu", "NWSMKRYBDHV-nwsmkrybdhv"};
        for (int c = 0; c < 5; c++)
            for (const char *p = classes[c]; *p; p++) nt[(uint8_t)*p] = (uint8_t)c;
        for (int i = 0; i < 26; i++) aa['A' + i] = aa['a' + i] = (uint8_t)i;
        aa[(uint8_t)'*'] = 26;
        aa[(uint8_t)'-'] = 27;
    }
};
const Tables &tables() {
    static const Tables t;
    return t;
}
}  // namespace

uint8_t code_of(int alphabet, uint8_t byte) {
    return alphabet == SMAFA_ALPHABET_AA ? tables().aa[byte] : tables().nt[byte];
}

const uint8_t *code_table(int alphabet) { return alphabet == SMAFA_ALPHABET_AA ? tables().aa : tables().nt; }

char letter_of(int alphabet, uint8_t code) {
    if (alphabet == SMAFA_ALPHABET_AA) return code < 26 ? (char)('A' + code) : code == 26 ? '*' : code == 27 ? '-' : '?';
    return code < 5 ? "ACGTN"[code] : '?';
}

const char *alphabet_noun(int alphabet) { return alphabet == SMAFA_ALPHABET_AA ? "amino acid" : "nucleotide"; }

}  // namespace smafa

extern "C" {

int smafa_encode(int alphabet, const uint8_t *ascii, uint64_t len, uint8_t *codes, uint64_t *bad_pos) try {
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return smafa::set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    if ((!ascii || !codes) && len) return smafa::set_error(SMAFA_ERR_INVALID, "smafa_encode: NULL argument");
    const uint8_t *t = alphabet == SMAFA_ALPHABET_AA ? smafa::tables().aa : smafa::tables().nt;
    for (uint64_t i = 0; i < len; i++) {
        const uint8_t c = t[ascii[i]];
        if (c == 255) {
            if (bad_pos) *bad_pos = i;
            return smafa::set_error(SMAFA_ERR_PANIC, "Byte %u cannot be interpreted as %s at position %llu", ascii[i],
                                    smafa::alphabet_noun(alphabet), (unsigned long long)i);
        }
        codes[i] = c;
    }
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_encode");
}

int smafa_decode(int alphabet, const uint8_t *codes, uint64_t len, char *out) try {
    if (alphabet != SMAFA_ALPHABET_NT && alphabet != SMAFA_ALPHABET_AA)
        return smafa::set_error(SMAFA_ERR_INVALID, "unknown alphabet %d", alphabet);
    for (uint64_t i = 0; i < len; i++) {
        const char c = smafa::letter_of(alphabet, codes[i]);
        if (c == '?') return smafa::set_error(SMAFA_ERR_PANIC, "Invalid character in query sequence: %u", codes[i]);
        out[i] = c;
    }
    return SMAFA_OK;
} catch (...) {
    return smafa::exception_code("smafa_decode");
}

}
