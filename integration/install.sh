#!/bin/sh
# Puts the MI355X modules into an UltraGrid source tree and patches its configure.ac (integration/ultragrid_mi355x.patch):
#   sh integration/install.sh <ultragrid-source-dir>
# then, in that tree:  ./autogen.sh --with-ug-mi355x=<prefix holding lib/libug_mi355x.so> && make
# Files keep their content; only their names follow the reference's layout (src/video_compress/<name>.cpp -> ultragrid_vcompress_<name>.so).
# The module sources include the C ABI as "../../include/ug_mi355x.h": all three destination directories are two levels below the
# tree's root, so the header goes to <tree>/include/.
set -e
UG=${1:?usage: install.sh <ultragrid-source-dir>}
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(dirname "$HERE"); M=$ROOT/ultragrid_amd/module
test -f "$UG/configure.ac" || { echo "$UG: no configure.ac there" >&2; exit 1; }
mkdir -p "$UG/include" "$UG/src/video_compress" "$UG/src/video_decompress" "$UG/src/libavcodec"
cp "$ROOT/include/ug_mi355x.h"               "$UG/include/ug_mi355x.h"
cp "$M/vcompress_dxt_mi355x.cpp"             "$UG/src/video_compress/dxt_mi355x.cpp"
cp "$M/vcompress_jpeg_mi355x.cpp"            "$UG/src/video_compress/jpeg_mi355x.cpp"
cp "$M/ug_codec_map.h" "$M/mi355x_frame_sharder.h" "$UG/src/video_compress/"
cp "$M/vdecompress_dxt_mi355x.c"             "$UG/src/video_decompress/dxt_mi355x.c"
cp "$M/vdecompress_jpeg_mi355x.c"            "$UG/src/video_decompress/jpeg_mi355x.c"
cp "$M/vdecompress_jpeg_to_dxt_mi355x.c"     "$UG/src/video_decompress/jpeg_to_dxt_mi355x.c"
cp "$M/mi355x_receiver.h"                    "$UG/src/video_decompress/"
cp "$M/lavc_conv_mi355x.cpp"                 "$UG/src/libavcodec/lavc_conv_mi355x.cpp"
if grep -q "found_ug_mi355x" "$UG/configure.ac"; then
        echo "configure.ac is patched already"
else
        patch -p1 -d "$UG" < "$HERE/ultragrid_mi355x.patch"
fi
echo "installed into $UG: 5 modules + the lavc hook; configure with --with-ug-mi355x=<prefix>"
