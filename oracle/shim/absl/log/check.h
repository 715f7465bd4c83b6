// Build shim (test infrastructure) for abseil CHECK/DCHECK macros. Not product code.
#pragma once
#include <cstdio>
#include <cstdlib>
#define TFC_SHIM_CHECK(cond)                                                   \
  do {                                                                         \
    if (!(cond)) {                                                             \
      std::fprintf(stderr, "CHECK failed: %s at %s:%d\n", #cond, __FILE__,     \
                   __LINE__);                                                  \
      std::abort();                                                            \
    }                                                                          \
  } while (0)
#ifndef CHECK
#define CHECK(c) TFC_SHIM_CHECK(c)
#define CHECK_EQ(a, b) TFC_SHIM_CHECK((a) == (b))
#define CHECK_NE(a, b) TFC_SHIM_CHECK((a) != (b))
#define CHECK_LT(a, b) TFC_SHIM_CHECK((a) < (b))
#define CHECK_LE(a, b) TFC_SHIM_CHECK((a) <= (b))
#define CHECK_GT(a, b) TFC_SHIM_CHECK((a) > (b))
#define CHECK_GE(a, b) TFC_SHIM_CHECK((a) >= (b))
#endif
#ifndef DCHECK
#ifdef TFC_SHIM_DCHECK_ON
#define DCHECK(c) TFC_SHIM_CHECK(c)
#define DCHECK_EQ(a, b) TFC_SHIM_CHECK((a) == (b))
#define DCHECK_NE(a, b) TFC_SHIM_CHECK((a) != (b))
#define DCHECK_LT(a, b) TFC_SHIM_CHECK((a) < (b))
#define DCHECK_LE(a, b) TFC_SHIM_CHECK((a) <= (b))
#define DCHECK_GT(a, b) TFC_SHIM_CHECK((a) > (b))
#define DCHECK_GE(a, b) TFC_SHIM_CHECK((a) >= (b))
#else
#define DCHECK(c) ((void)0)
#define DCHECK_EQ(a, b) ((void)0)
#define DCHECK_NE(a, b) ((void)0)
#define DCHECK_LT(a, b) ((void)0)
#define DCHECK_LE(a, b) ((void)0)
#define DCHECK_GT(a, b) ((void)0)
#define DCHECK_GE(a, b) ((void)0)
#endif
#endif
