// csrc/ssd_codeobj.S -- the device code of ssd_kernels.hip once more, as plain data: the clang offload bundle that hipcc
// put into ssd_kernels.o's .hip_fatbin section (extracted by the Makefile).  ssd_aql.hip loads the gfx950 code object in it
// through the HSA executable API for the library's own AQL dispatches; the HIP runtime keeps using the original section.
    .section .rodata.ssd_codeobj, "a", @progbits
    .balign 4096
    .globl ssd_kernels_bundle
    .globl ssd_kernels_bundle_end
ssd_kernels_bundle:
    .incbin "ssd_kernels.bundle"
ssd_kernels_bundle_end:
    .byte 0
    .section .note.GNU-stack, "", @progbits
