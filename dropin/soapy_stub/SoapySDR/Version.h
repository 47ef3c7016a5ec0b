/* Stand-in for <SoapySDR/Version.h>.
 * SoapySDR (rx_tools' only third-party dependency, its CMakeLists.txt:17) may be absent on the build machine -- it is in this
 * project's image.  rx_tools' DSP never touches it; these headers plus soapy_fake.c (a capture-replay device) let rx_tools'
 * src/*.c compile unmodified and run from a file: dropin/Makefile with SOAPY=stub, and oracle/Makefile for the reference-built
 * checker objects.  Declarations follow the public SoapySDR 0.8 C API. */
#pragma once
#define SOAPY_SDR_API_VERSION 0x00080000
#define SOAPY_SDR_ABI_VERSION "0.8"
