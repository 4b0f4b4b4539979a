/* Embeds data/dictionary.bin (RFC 7932 Appendix A, CRC-32 0x5136cb04) into the oracle library.
 * TEST INFRASTRUCTURE ONLY -- see brotli_oracle.c. */
#ifndef DICT_PATH
#error "DICT_PATH must point at rust-brotli-decompressor_amd/data/dictionary.bin"
#endif
__asm__(".section .rodata\n"
        ".balign 64\n"
        ".global brotli_oracle_dictionary\n"
        ".type brotli_oracle_dictionary, @object\n"
        "brotli_oracle_dictionary:\n"
        ".incbin \"" DICT_PATH "\"\n"
        ".size brotli_oracle_dictionary, .-brotli_oracle_dictionary\n"
        ".previous\n");
