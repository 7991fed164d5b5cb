#!/usr/bin/env python3
"""Regenerates integration/ultragrid_mi355x.patch from the reference's configure.ac (default /root/reference).

The patch is what a maintainer applies to an UltraGrid checkout (after integration/install.sh has copied the module sources in):
four insertions into configure.ac, nothing removed --
  1. detection of libug_mi355x (--with-ug-mi355x=<prefix>) in front of the Libav section (that section needs the answer),
  2. the lavc conversion hook: HAVE_LAVC_CUDA_CONV with src/libavcodec/lavc_conv_mi355x.o in place of the two stubbed *_cuda.o objects
     (configure.ac:2056-2069 of the reference; the hook's declarations are the reference's own *_cuda.h headers),
  3. the five modules through the reference's add_module helper (configure.ac:243-259), behind the CUDA DXT section,
  4. a line in the summary table.
Only `diff -u` context lines of the reference appear in the patch; no reference source is copied into this repository.
"""
import difflib, os, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

DETECT = '''# -------------------------------------------------------------------------------------------------
# MI355X kernel library (libug_mi355x: pixel formats, DXT, JPEG on AMD CDNA4 through a C ABI)
# -------------------------------------------------------------------------------------------------
ug_mi355x=no
found_ug_mi355x=no
AC_ARG_WITH(ug-mi355x,
        AS_HELP_STRING([--with-ug-mi355x=DIR], [prefix of libug_mi355x (DIR/lib/libug_mi355x.so; the header travels with the module sources, include/ug_mi355x.h); default is to look in the system paths]),
        [ug_mi355x_req=yes; UG_MI355X_PREFIX=$withval],
        [ug_mi355x_req=$build_default; UG_MI355X_PREFIX=])
if test "${ug_mi355x_req?}" != no; then
        ug_mi355x_saved_LIBS=$LIBS
        ug_mi355x_saved_LDFLAGS=$LDFLAGS
        if test -n "$UG_MI355X_PREFIX" && test "$UG_MI355X_PREFIX" != yes; then
                UG_MI355X_LIB="-L$UG_MI355X_PREFIX/lib -Wl,-rpath,$UG_MI355X_PREFIX/lib -lug_mi355x"
                LDFLAGS="$LDFLAGS -L$UG_MI355X_PREFIX/lib"
        else
                UG_MI355X_LIB="-lug_mi355x"
        fi
        AC_CHECK_LIB(ug_mi355x, ug_hip_abi_version, [found_ug_mi355x=yes], [found_ug_mi355x=no])
        LIBS=$ug_mi355x_saved_LIBS
        LDFLAGS=$ug_mi355x_saved_LDFLAGS
fi

'''

LAVC = '''        if test "$lavc_cuda" != yes && test "$found_ug_mi355x" = yes; then
                # the same hook (to/from_lavc_vid_conv_cuda.h), answered by libug_mi355x instead of the stubbed .cu files
                AC_DEFINE([HAVE_LAVC_CUDA_CONV], [1], [Build with lavc CUDA conversions])
                to_lavc_cuda_obj=src/libavcodec/lavc_conv_mi355x.o
                LIBAVCODEC_VIDEO="$LIBAVCODEC_VIDEO $to_lavc_cuda_obj"
                LIBAVCODEC_LIBS="$LIBAVCODEC_LIBS $UG_MI355X_LIB"
        fi
'''

MODULES = '''
# -------------------------------------------------------------------------------------------------
# MI355X DXT / JPEG compression and decompression (-c dxt, -c jpeg; needs no CUDA, no GL context)
# -------------------------------------------------------------------------------------------------
if test "${found_ug_mi355x?}" = yes
then
        ug_mi355x=yes
        add_module vcompress_dxt "src/video_compress/dxt_mi355x.o" "$UG_MI355X_LIB"
        add_module vcompress_jpeg "src/video_compress/jpeg_mi355x.o" "$UG_MI355X_LIB"
        add_module vdecompress_dxt_mi355x "src/video_decompress/dxt_mi355x.o" "$UG_MI355X_LIB"
        add_module vdecompress_jpeg_mi355x "src/video_decompress/jpeg_mi355x.o" "$UG_MI355X_LIB"
        add_module vdecompress_jpeg_to_dxt_mi355x "src/video_decompress/jpeg_to_dxt_mi355x.o" "$UG_MI355X_LIB"
fi

ENSURE_FEATURE_PRESENT([$ug_mi355x_req], [$ug_mi355x], [libug_mi355x not found])
'''

SUMMARY = 'add_column "MI355X DXT/JPEG" "${ug_mi355x?}"\n'


def insert_before(lines, needle, text, nth=0):
    idx = [i for i, l in enumerate(lines) if l.rstrip("\n") == needle]
    if len(idx) <= nth:
        raise SystemExit(f"anchor not found in configure.ac: {needle!r}")
    i = idx[nth]
    return lines[:i] + text.splitlines(keepends=True) + lines[i:]


def insert_after(lines, needle, text):
    idx = [i for i, l in enumerate(lines) if l.rstrip("\n") == needle]
    if len(idx) != 1:
        raise SystemExit(f"anchor not found exactly once in configure.ac: {needle!r}")
    i = idx[0] + 1
    return lines[:i] + text.splitlines(keepends=True) + lines[i:]


def main():
    orig = open(os.path.join(REF, "configure.ac")).read().splitlines(keepends=True)
    new = list(orig)
    # 1. in front of the banner line that opens the "# Libav" section
    libav = [i for i, l in enumerate(new) if l.rstrip("\n") == "# Libav"]
    if len(libav) != 1:
        raise SystemExit("the '# Libav' section header was not found exactly once")
    i = libav[0] - 1   # the dashed line above it
    new = new[:i] + DETECT.splitlines(keepends=True) + new[i:]
    # 2. behind the `if test "$lavc_cuda" = yes; then ... fi` block = in front of the HAVE_LAVC define
    new = insert_before(new, "        AC_DEFINE([HAVE_LAVC], [1], [Build with LAVC support])", LAVC)
    # 3. behind the CUDA DXT section
    new = insert_after(new, "ENSURE_FEATURE_PRESENT([$cuda_dxt_req], [$cuda_dxt], [CUDA DXT not found])", MODULES)
    # 4. summary table, behind "Lavc ..." keeps the list alphabetical enough: in front of OpenAPV
    new = insert_before(new, 'add_column "OpenAPV" "${openapv?}"', SUMMARY)
    diff = difflib.unified_diff(orig, new, "a/configure.ac", "b/configure.ac", n=3)
    out = os.path.join(HERE, "ultragrid_mi355x.patch")
    with open(out, "w") as f:
        f.writelines(diff)
    print(f"wrote {out}: {sum(1 for l in open(out) if l.startswith('+') and not l.startswith('+++'))} lines added")


if __name__ == "__main__":
    main()
