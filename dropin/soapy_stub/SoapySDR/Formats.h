/* Stand-in for <SoapySDR/Formats.h> -- the capture-replay stand-in (see Version.h). */
#pragma once
#include <stddef.h>
#define SOAPY_SDR_CF64 "CF64"
#define SOAPY_SDR_CF32 "CF32"
#define SOAPY_SDR_CS32 "CS32"
#define SOAPY_SDR_CU32 "CU32"
#define SOAPY_SDR_CS16 "CS16"
#define SOAPY_SDR_CU16 "CU16"
#define SOAPY_SDR_CS12 "CS12"
#define SOAPY_SDR_CU12 "CU12"
#define SOAPY_SDR_CS8  "CS8"
#define SOAPY_SDR_CU8  "CU8"
#define SOAPY_SDR_CS4  "CS4"
#define SOAPY_SDR_CU4  "CU4"
#define SOAPY_SDR_F64  "F64"
#define SOAPY_SDR_F32  "F32"
#define SOAPY_SDR_S32  "S32"
#define SOAPY_SDR_U32  "U32"
#define SOAPY_SDR_S16  "S16"
#define SOAPY_SDR_U16  "U16"
#define SOAPY_SDR_S8   "S8"
#define SOAPY_SDR_U8   "U8"
#ifdef __cplusplus
extern "C" {
#endif
size_t SoapySDR_formatToSize(const char *format);
#ifdef __cplusplus
}
#endif
