// oracle/ref_lab_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
// C entry point around the reference's OWN colour conversion: the header is included from where it lies under the
// reference tree (-I$(REF)/simplestereo/headers, oracle/Makefile target `ref`), never copied.  Lets
// tests/test_oracle_golden.py pin oracle_bgr2lab (oracle_passive.c) against ColorConversion::ImageFromBGR2Lab
// (headers/colorconversion.hpp:81-86) directly instead of only through the disparity maps.
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <Python.h>
#include <numpy/arrayobject.h>
#include "colorconversion.hpp"

extern "C" void ref_bgr2lab(unsigned char *bgr, double *lab, int width, int height)
{
    ColorConversion cc;
    cc.ImageFromBGR2Lab(bgr, lab, width, height);
}
