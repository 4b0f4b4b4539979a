/* Embeds data/dictionary.bin -- the RFC 7932 Appendix A static dictionary (122784 bytes, CRC-32 0x5136cb04,
 * the same bytes as the reference's src/dictionary/mod.rs:18-7692) -- into libbrotli_decompressor.so.
 * The host uploads it to each device once (brotli_capi.cpp: device_dictionary). */
#ifndef DICT_PATH
#error "DICT_PATH must point at data/dictionary.bin"
#endif
__asm__(".section .rodata\n"
        ".balign 64\n"
        ".global brotli_amd_dictionary\n"
        ".type brotli_amd_dictionary, @object\n"
        "brotli_amd_dictionary:\n"
        ".incbin \"" DICT_PATH "\"\n"
        ".size brotli_amd_dictionary, .-brotli_amd_dictionary\n"
        ".previous\n");
