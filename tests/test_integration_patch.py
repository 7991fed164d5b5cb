"""integration/: the reference-side binding as something a maintainer can apply -- install.sh copies the module sources into an UltraGrid tree
under the names the reference's build expects and patches configure.ac (add_module, configure.ac:243-259; the lavc hook, :2056-2069).
Checked here against the reference's own configure.ac and headers (no autoconf in this image: the patched script is not executed)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "configure.ac")), reason="needs the reference tree (not on the GPU box)")


def _tree(tmp_path):
    ug = tmp_path / "UltraGrid"
    ug.mkdir()
    shutil.copy(os.path.join(REF, "configure.ac"), ug / "configure.ac")
    return ug


def test_committed_patch_is_what_the_generator_makes(tmp_path):
    gen = tmp_path / "integration"
    gen.mkdir()
    shutil.copy(os.path.join(ROOT, "integration", "make_patch.py"), gen / "make_patch.py")
    subprocess.run([sys.executable, str(gen / "make_patch.py"), REF], check=True, capture_output=True)
    assert (gen / "ultragrid_mi355x.patch").read_text() == open(os.path.join(ROOT, "integration", "ultragrid_mi355x.patch")).read()


def test_install_places_the_sources_and_patches_configure(tmp_path):
    ug = _tree(tmp_path)
    before = (ug / "configure.ac").read_text()
    r = subprocess.run(["sh", os.path.join(ROOT, "integration", "install.sh"), str(ug)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    after = (ug / "configure.ac").read_text()
    # additions only: every line of the original is still there, in order
    it = iter(after.splitlines())
    assert all(any(line == other for other in it) for line in before.splitlines())
    for needle in ("AC_CHECK_LIB(ug_mi355x, ug_hip_abi_version", 'add_module vcompress_dxt "src/video_compress/dxt_mi355x.o"',
                   'add_module vcompress_jpeg "src/video_compress/jpeg_mi355x.o"', "add_module vdecompress_dxt_mi355x", "add_module vdecompress_jpeg_mi355x",
                   "add_module vdecompress_jpeg_to_dxt_mi355x", "to_lavc_cuda_obj=src/libavcodec/lavc_conv_mi355x.o", 'add_column "MI355X DXT/JPEG"'):
        assert needle in after, needle
    # the detection comes before the Libav section that asks for its answer, the modules after add_module is defined
    assert after.index("found_ug_mi355x=no") < after.index("# Libav") < after.index('test "$found_ug_mi355x" = yes')
    assert after.index("add_module() {") < after.index("add_module vcompress_dxt ")
    for f in ("include/ug_mi355x.h", "src/video_compress/dxt_mi355x.cpp", "src/video_compress/jpeg_mi355x.cpp", "src/video_compress/ug_codec_map.h",
              "src/video_compress/mi355x_frame_sharder.h", "src/video_decompress/dxt_mi355x.c", "src/video_decompress/jpeg_mi355x.c",
              "src/video_decompress/jpeg_to_dxt_mi355x.c", "src/video_decompress/mi355x_receiver.h", "src/libavcodec/lavc_conv_mi355x.cpp"):
        assert (ug / f).is_file(), f
    # a second run leaves the tree as it is
    r = subprocess.run(["sh", os.path.join(ROOT, "integration", "install.sh"), str(ug)], capture_output=True, text=True)
    assert r.returncode == 0 and "patched already" in r.stdout
    assert (ug / "configure.ac").read_text() == after


@pytest.mark.parametrize("src,std", [("src/video_compress/dxt_mi355x.cpp", "gnu++20"), ("src/video_compress/jpeg_mi355x.cpp", "gnu++20"),
                                     ("src/video_decompress/dxt_mi355x.c", "gnu2x"), ("src/video_decompress/jpeg_mi355x.c", "gnu2x"),
                                     ("src/video_decompress/jpeg_to_dxt_mi355x.c", "gnu2x")])
def test_installed_sources_compile_where_they_were_put(tmp_path, src, std):
    """The relative include of the C ABI ("../../include/ug_mi355x.h") and the module-local headers resolve in the destination layout; the
    UltraGrid headers come from the reference tree."""
    ug = _tree(tmp_path)
    subprocess.run(["sh", os.path.join(ROOT, "integration", "install.sh"), str(ug)], check=True, capture_output=True)
    cc = ["g++", "-std=" + std] if src.endswith(".cpp") else ["gcc", "-std=" + std]
    r = subprocess.run(cc + ["-fsyntax-only", "-D_GNU_SOURCE", "-msse4.1", "-I", os.path.join(REF, "src"), str(ug / src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_library_installs_into_the_prefix_the_patch_looks_in(tmp_path):
    """`make -C ultragrid_amd/csrc install PREFIX=<dir>` lays out <dir>/lib/libug_mi355x.so + <dir>/include/ug_mi355x.h -- what the patched
    configure.ac's --with-ug-mi355x=<dir> puts on the link line (-L<dir>/lib -lug_mi355x); the installed library exports the symbol AC_CHECK_LIB asks for."""
    so = os.path.join(ROOT, "ultragrid_amd", "libug_mi355x.so")
    if not os.path.exists(so):
        pytest.skip("library not built yet (build() runs first in the driver's order)")
    prefix = tmp_path / "prefix"
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "ultragrid_amd", "csrc"), "install", f"PREFIX={prefix}", "VARIANT=0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (prefix / "include" / "ug_mi355x.h").is_file()
    nm = subprocess.run(["nm", "-D", "--defined-only", str(prefix / "lib" / "libug_mi355x.so")], capture_output=True, text=True, check=True).stdout
    assert " T ug_hip_abi_version" in nm
    patch = open(os.path.join(ROOT, "integration", "ultragrid_mi355x.patch")).read()
    assert "AC_CHECK_LIB(ug_mi355x, ug_hip_abi_version" in patch and "-L$UG_MI355X_PREFIX/lib" in patch
