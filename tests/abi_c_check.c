/* Compiled as C99 by tests/test_abi.py: the public header must be plain C, and a C host must be able to link
 * libsmafa_amd.so and walk the host-only part of the ABI (encode, DB file round trip, error channel). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smafa_amd.h"

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    uint8_t codes[8];
    uint64_t bad = 0;
    if (smafa_encode(SMAFA_ALPHABET_NT, (const uint8_t *)"ACGTNRYU", 8, codes, &bad) != SMAFA_OK) return 3;
    const uint8_t want[8] = {0, 1, 2, 3, 4, 4, 4, 3};
    if (memcmp(codes, want, 8)) return 4;
    if (smafa_encode(SMAFA_ALPHABET_NT, (const uint8_t *)"ACE", 3, codes, &bad) != SMAFA_ERR_PANIC || bad != 2) return 5;
    if (!strstr(smafa_last_error(), "Byte 69 cannot be interpreted as nucleotide")) return 6;
    if (smafa_dbfile_write(argv[1], SMAFA_ALPHABET_NT, want, 2, 4) != SMAFA_OK) return 7;
    int alphabet = -1;
    uint8_t *back = NULL;
    uint64_t n = 0;
    uint32_t len = 0;
    if (smafa_dbfile_read(argv[1], &alphabet, &back, &n, &len) != SMAFA_OK) return 8;
    if (alphabet != SMAFA_ALPHABET_NT || n != 2 || len != 4 || memcmp(back, want, 8)) return 9;
    smafa_free(back);
    char text[9] = {0};
    if (smafa_decode(SMAFA_ALPHABET_NT, want, 8, text) != SMAFA_OK || strcmp(text, "ACGTNNNT")) return 10;
    smafa_db *db = NULL;
    int rc = smafa_db_create(&db, 0, SMAFA_ALPHABET_NT, 4);
    if (smafa_device_count() == 0) {
        if (rc != SMAFA_ERR_DEVICE || db != NULL) return 11; /* no silent fallback */
    } else {
        if (rc != SMAFA_OK) return 12;
        smafa_db_destroy(db);
    }
    printf("abi ok, devices=%d\n", smafa_device_count());
    return 0;
}
