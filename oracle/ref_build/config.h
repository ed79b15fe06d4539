/* Hand-written build configuration for compiling the reference's HWLM hot-path
 * sources in place (names from /root/reference/cmake/config.h.in).
 * Test/oracle infrastructure only -- never linked into the product library. */
#ifndef CONFIG_H_
#define CONFIG_H_
#define ARCH_64_BIT
#define ARCH_X86_64
#define HAVE_C_X86INTRIN_H
#define HAVE_CXX_X86INTRIN_H
#define HAVE_CC_BUILTIN_ASSUME_ALIGNED
#define HAVE_CXX_BUILTIN_ASSUME_ALIGNED
#define HAVE_POSIX_MEMALIGN
#define HAVE_UNISTD_H
#define HAVE__BUILTIN_CONSTANT_P
#define HS_OPTIMIZE
/* fdr_compile.cpp and the driver are built with -DHSREF_DEV_HOOKS so that the
 * reference's unit-test hook fdrBuildProtoHinted (fdr_compile.cpp:900-911) exists. */
#ifndef HSREF_DEV_HOOKS
#define RELEASE_BUILD
#endif
#endif
