/* Stand-in for <SoapySDR/Version.h> -- TEST INFRASTRUCTURE ONLY.
 * SoapySDR (the reference's only third-party dependency, CMakeLists.txt:17) is
 * not installed in this image.  The reference's DSP never touches it; these
 * headers exist so /root/reference/src/*.c can be compiled unmodified into
 * oracle/_ref/.  Declarations follow the public SoapySDR 0.8 C API. */
#pragma once
#define SOAPY_SDR_API_VERSION 0x00080000
#define SOAPY_SDR_ABI_VERSION "0.8"
