/* glsl_ref.c -- runs the reference's OWN GLSL encoders (dxt_compress/compress_dxt5ycocg_fp.glsl, compress_dxt1_fp.glsl,
 * yuv422_to_yuv444.glsl, compress_vp.glsl) on the CPU with Mesa's llvmpipe, headless, so that oracle/dxt_oracle.c can be
 * pinned to the reference implementation itself (the shaders are the normative DXT encoders and there is
 * no GPU / GL context in the build container).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing of the reference is copied: the shader sources are read at run time from
 * <refdir>/dxt_compress/ and get the same three-line preamble the reference's build puts in front of them
 * (Makefile.in:366-414: "#version 140", "#define legacy 0", "#define FORMAT_YUV n").  The GL calls restate what
 * dxt_compress/dxt_encoder.c does (texture formats :219-220,293-296,362-364; uniforms :386-394; quad :66-73; 4:2:2 pre-pass
 * :482-533; read-back :670-671).
 *
 * The context comes from Mesa's software rasteriser loaded through the DRI swrast interface (GL/internal/dri_interface.h):
 * no X server, no EGL, no GLEW.
 *
 * usage: glsl_ref <refdir> <dxt5|dxt1|dxt1yuv> <rgb|rgba|yuv444|uyvy> <width> <height> <in.raw> <out.bin>
 *        glsl_ref <refdir> <dec5|dec1|dec1yuv> rgba <width> <height> <in.dxt> <out.rgba>   (the GL decoder + display shaders)
 *        glsl_ref <refdir> rgba2uyvy rgba <width> <height> <in.rgba> <out.uyvy>            (rgba_to_yuv422.glsl alone)
 *        (dxt1yuv = DXT_TYPE_DXT1_YUV: the DXT1 shader WITHOUT the YUV->RGB step on Y,U,V samples, dxt_encoder.c:320-323)
 *        (yuv444 = DXT_FORMAT_YUV: 4 bytes per pixel Y U V x, as the reference's RGBA upload path takes it)
 */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <GL/gl.h>
#include <GL/glext.h>
#include <GL/internal/dri_interface.h>

static void get_drawable_info(__DRIdrawable *d, int *x, int *y, int *w, int *h, void *p) { (void) d; (void) p; *x = *y = 0; *w = *h = 16; }
static void put_image(__DRIdrawable *d, int op, int x, int y, int w, int h, char *data, void *p) { (void) d; (void) op; (void) x; (void) y; (void) w; (void) h; (void) data; (void) p; }
static void get_image(__DRIdrawable *d, int x, int y, int w, int h, char *data, void *p) { (void) d; (void) x; (void) y; (void) p; memset(data, 0, (size_t) w * h * 4); }
static const __DRIswrastLoaderExtension swrast_loader = { .base = { __DRI_SWRAST_LOADER, 1 }, .getDrawableInfo = get_drawable_info, .putImage = put_image, .getImage = get_image };
static const __DRIextension *loader_exts[] = { &swrast_loader.base, NULL };

static void *(*gpa)(const char *);
#define GLF(ret, name, ...) static ret (*name)(__VA_ARGS__)
GLF(const GLubyte *, p_glGetString, GLenum);
GLF(GLuint, p_glCreateShader, GLenum);
GLF(void, p_glShaderSource, GLuint, GLsizei, const GLchar *const *, const GLint *);
GLF(void, p_glCompileShader, GLuint);
GLF(void, p_glGetShaderiv, GLuint, GLenum, GLint *);
GLF(void, p_glGetShaderInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(GLuint, p_glCreateProgram, void);
GLF(void, p_glAttachShader, GLuint, GLuint);
GLF(void, p_glLinkProgram, GLuint);
GLF(void, p_glGetProgramiv, GLuint, GLenum, GLint *);
GLF(void, p_glGetProgramInfoLog, GLuint, GLsizei, GLsizei *, GLchar *);
GLF(void, p_glUseProgram, GLuint);
GLF(GLint, p_glGetUniformLocation, GLuint, const GLchar *);
GLF(void, p_glUniform1i, GLint, GLint);
GLF(void, p_glUniform1f, GLint, GLfloat);
GLF(void, p_glUniform2f, GLint, GLfloat, GLfloat);
GLF(void, p_glGenTextures, GLsizei, GLuint *);
GLF(void, p_glBindTexture, GLenum, GLuint);
GLF(void, p_glTexParameteri, GLenum, GLenum, GLint);
GLF(void, p_glTexImage2D, GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void *);
GLF(void, p_glTexSubImage2D, GLenum, GLint, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, const void *);
GLF(void, p_glGenFramebuffers, GLsizei, GLuint *);
GLF(void, p_glBindFramebuffer, GLenum, GLuint);
GLF(void, p_glFramebufferTexture2D, GLenum, GLenum, GLenum, GLuint, GLint);
GLF(GLenum, p_glCheckFramebufferStatus, GLenum);
GLF(void, p_glViewport, GLint, GLint, GLsizei, GLsizei);
GLF(void, p_glDisable, GLenum);
GLF(void, p_glGenVertexArrays, GLsizei, GLuint *);
GLF(void, p_glBindVertexArray, GLuint);
GLF(void, p_glGenBuffers, GLsizei, GLuint *);
GLF(void, p_glBindBuffer, GLenum, GLuint);
GLF(void, p_glBufferData, GLenum, GLsizeiptr, const void *, GLenum);
GLF(GLint, p_glGetAttribLocation, GLuint, const GLchar *);
GLF(void, p_glVertexAttribPointer, GLuint, GLint, GLenum, GLboolean, GLsizei, const void *);
GLF(void, p_glEnableVertexAttribArray, GLuint);
GLF(void, p_glDrawArrays, GLenum, GLint, GLsizei);
GLF(void, p_glDrawBuffer, GLenum);
GLF(void, p_glReadBuffer, GLenum);
GLF(void, p_glReadPixels, GLint, GLint, GLsizei, GLsizei, GLenum, GLenum, void *);
GLF(void, p_glClearColor, GLfloat, GLfloat, GLfloat, GLfloat);
GLF(void, p_glClear, GLbitfield);
GLF(void, p_glFinish, void);
GLF(GLenum, p_glGetError, void);
GLF(void, p_glCompressedTexImage2D, GLenum, GLint, GLenum, GLsizei, GLsizei, GLint, GLsizei, const void *);
#define LOAD(n) do { *(void **) &p_##n = gpa(#n); if (!p_##n) { fprintf(stderr, "missing %s\n", #n); return 2; } } while (0)

static int make_context(void)
{
        void *h = dlopen("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("swrast_dri.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "swrast_dri.so: %s\n", dlerror()); return 2; }
        const __DRIextension **(*get_ext)(void) = (const __DRIextension **(*)(void)) dlsym(h, "__driDriverGetExtensions_swrast");
        if (!get_ext) { fprintf(stderr, "no __driDriverGetExtensions_swrast\n"); return 2; }
        const __DRIextension **exts = get_ext();
        const __DRIcoreExtension *core = NULL;
        const __DRIswrastExtension *sw = NULL;
        for (int i = 0; exts[i]; i++) {
                if (!strcmp(exts[i]->name, __DRI_CORE)) core = (const __DRIcoreExtension *) exts[i];
                if (!strcmp(exts[i]->name, __DRI_SWRAST)) sw = (const __DRIswrastExtension *) exts[i];
        }
        if (!core || !sw || sw->base.version < 4) { fprintf(stderr, "DRI core / swrast(v4) extension missing\n"); return 2; }
        const __DRIconfig **configs = NULL;
        __DRIscreen *scr = sw->createNewScreen2(0, loader_exts, exts, &configs, NULL);
        if (!scr || !configs || !configs[0]) { fprintf(stderr, "createNewScreen2 failed\n"); return 2; }
        unsigned err = 0;
        uint32_t attribs[] = { __DRI_CTX_ATTRIB_MAJOR_VERSION, 3, __DRI_CTX_ATTRIB_MINOR_VERSION, 1 };
        __DRIcontext *ctx = sw->createContextAttribs(scr, __DRI_API_OPENGL, configs[0], NULL, 2, attribs, &err, NULL);
        __DRIdrawable *dr = ctx ? sw->createNewDrawable(scr, configs[0], NULL) : NULL;
        if (!ctx || !dr || !core->bindContext(ctx, dr, dr)) { fprintf(stderr, "context creation failed (err %u)\n", err); return 2; }
        void *glapi = dlopen("libglapi.so.0", RTLD_NOW | RTLD_GLOBAL);
        gpa = glapi ? (void *(*)(const char *)) dlsym(glapi, "_glapi_get_proc_address") : NULL;
        if (!gpa) { fprintf(stderr, "libglapi: no _glapi_get_proc_address\n"); return 2; }
        LOAD(glGetString); LOAD(glCreateShader); LOAD(glShaderSource); LOAD(glCompileShader); LOAD(glGetShaderiv); LOAD(glGetShaderInfoLog);
        LOAD(glCreateProgram); LOAD(glAttachShader); LOAD(glLinkProgram); LOAD(glGetProgramiv); LOAD(glGetProgramInfoLog); LOAD(glUseProgram);
        LOAD(glGetUniformLocation); LOAD(glUniform1i); LOAD(glUniform1f); LOAD(glUniform2f); LOAD(glGenTextures); LOAD(glBindTexture);
        LOAD(glTexParameteri); LOAD(glTexImage2D); LOAD(glTexSubImage2D); LOAD(glGenFramebuffers); LOAD(glBindFramebuffer);
        LOAD(glFramebufferTexture2D); LOAD(glCheckFramebufferStatus); LOAD(glViewport); LOAD(glDisable); LOAD(glGenVertexArrays);
        LOAD(glBindVertexArray); LOAD(glGenBuffers); LOAD(glBindBuffer); LOAD(glBufferData); LOAD(glGetAttribLocation);
        LOAD(glVertexAttribPointer); LOAD(glEnableVertexAttribArray); LOAD(glDrawArrays); LOAD(glDrawBuffer); LOAD(glReadBuffer);
        LOAD(glReadPixels); LOAD(glClearColor); LOAD(glClear); LOAD(glFinish); LOAD(glGetError); LOAD(glCompressedTexImage2D);
        return 0;
}

static char *slurp(const char *dir, const char *name)
{
        char path[4096];
        snprintf(path, sizeof path, "%s/dxt_compress/%s", dir, name);
        FILE *f = fopen(path, "rb");
        if (!f) { perror(path); exit(2); }
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        rewind(f);
        char *s = (char *) malloc(n + 1);
        if (fread(s, 1, n, f) != (size_t) n) { perror(path); exit(2); }
        s[n] = 0;
        fclose(f);
        return s;
}

/* preamble as generated by Makefile.in:366-414, then the file */
static GLuint compile(GLenum type, const char *dir, const char *file, int with_format, int format_yuv)
{
        char pre[128];
        snprintf(pre, sizeof pre, with_format ? "#version 140\n#define legacy 0\n#define FORMAT_YUV %d\n" : "#version 140\n#define legacy 0\n", format_yuv);
        char *body = slurp(dir, file);
        const GLchar *src[2] = { pre, body };
        GLuint sh = p_glCreateShader(type);
        p_glShaderSource(sh, 2, src, NULL);
        p_glCompileShader(sh);
        GLint ok = 0;
        p_glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
        if (!ok) {
                char log[8192];
                p_glGetShaderInfoLog(sh, sizeof log, NULL, log);
                fprintf(stderr, "%s: %s\n", file, log);
                exit(2);
        }
        free(body);
        return sh;
}

static GLuint link_program(GLuint vs, GLuint fs)
{
        GLuint p = p_glCreateProgram();
        p_glAttachShader(p, fs);
        p_glAttachShader(p, vs);
        p_glLinkProgram(p);
        GLint ok = 0;
        p_glGetProgramiv(p, GL_LINK_STATUS, &ok);
        if (!ok) {
                char log[8192];
                p_glGetProgramInfoLog(p, sizeof log, NULL, log);
                fprintf(stderr, "link: %s\n", log);
                exit(2);
        }
        return p;
}

static const GLfloat points[] = { -1, -1, 0, 1, 1, -1, 0, 1, -1, 1, 0, 1, /* second triangle */ 1, -1, 0, 1, -1, 1, 0, 1, 1, 1, 0, 1 }; /* dxt_encoder.c:66-73 */

static GLuint make_vao(GLuint program)
{
        GLuint vao, vbo;
        p_glGenVertexArrays(1, &vao);
        p_glBindVertexArray(vao);
        const GLint loc = p_glGetAttribLocation(program, "position");
        p_glGenBuffers(1, &vbo);
        p_glBindBuffer(GL_ARRAY_BUFFER, vbo);
        p_glBufferData(GL_ARRAY_BUFFER, sizeof points, points, GL_STATIC_DRAW);
        p_glVertexAttribPointer((GLuint) loc, 4, GL_FLOAT, GL_FALSE, 0, 0);
        p_glEnableVertexAttribArray((GLuint) loc);
        p_glBindVertexArray(0);
        return vao;
}

static void tex_params(void)
{
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
        p_glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);
}

/* dxt_decoder_create / dxt_decoder_decompress (dxt_compress/dxt_decoder.c:126-330,368-470): S3TC texture fetched by the fixed-function
 * sampler, display shader, RGBA8 framebuffer, read back as bytes R,G,B,A */
static int run_decode(const char *dir, const char *mode, int w, int h, const char *in_path, const char *out_path)
{
        const int dxt5 = !strcmp(mode, "dec5"), yuv = !strcmp(mode, "dec1yuv");
        const size_t in_len = (size_t) ((w + 3) / 4 * 4) * ((h + 3) / 4 * 4) / (dxt5 ? 1 : 2);
        uint8_t *in = (uint8_t *) malloc(in_len);
        FILE *f = fopen(in_path, "rb");
        if (!f || fread(in, 1, in_len, f) != in_len) { fprintf(stderr, "cannot read %zu bytes from %s\n", in_len, in_path); return 1; }
        fclose(f);
        if (make_context()) return 2;
        GLuint ctex, fbo, otex;
        p_glGenTextures(1, &ctex);
        p_glBindTexture(GL_TEXTURE_2D, ctex);
        tex_params();
        p_glCompressedTexImage2D(GL_TEXTURE_2D, 0, dxt5 ? GL_COMPRESSED_RGBA_S3TC_DXT5_EXT : GL_COMPRESSED_RGB_S3TC_DXT1_EXT, w, h, 0, (GLsizei) in_len, in);
        p_glGenFramebuffers(1, &fbo);
        p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
        p_glGenTextures(1, &otex);
        p_glBindTexture(GL_TEXTURE_2D, otex);
        tex_params();
        p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, w, h, 0, GL_RGBA, GL_UNSIGNED_INT_8_8_8_8_REV, 0);
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, otex, 0);
        if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fprintf(stderr, "framebuffer incomplete\n"); return 2; }
        const GLuint prog = link_program(compile(GL_VERTEX_SHADER, dir, "compress_vp.glsl", 0, 0),
                                         compile(GL_FRAGMENT_SHADER, dir, dxt5 ? "display_dxt5ycocg_fp.glsl" : (yuv ? "display_dxt1_yuv_fp.glsl" : "display_fp.glsl"), 0, 0));
        const GLuint vao = make_vao(prog);
        p_glViewport(0, 0, w, h);
        p_glDisable(GL_DEPTH_TEST);
        p_glUseProgram(prog);
        p_glUniform1i(p_glGetUniformLocation(prog, yuv ? "yuvtex" : "_image"), 0);
        p_glBindTexture(GL_TEXTURE_2D, ctex);
        p_glBindVertexArray(vao);
        p_glDrawArrays(GL_TRIANGLES, 0, 6);
        p_glBindVertexArray(0);
        p_glReadBuffer(GL_COLOR_ATTACHMENT0);
        uint8_t *out = (uint8_t *) calloc(4, (size_t) w * h);
        p_glReadPixels(0, 0, w, h, GL_RGBA, GL_UNSIGNED_INT_8_8_8_8_REV, out);
        p_glFinish();
        const GLenum e = p_glGetError();
        if (e != GL_NO_ERROR) { fprintf(stderr, "GL error 0x%x\n", e); return 2; }
        f = fopen(out_path, "wb");
        if (!f || fwrite(out, 4, (size_t) w * h, f) != (size_t) w * h) { perror(out_path); return 1; }
        fclose(f);
        return 0;
}

/* the RGBA -> 4:2:2 pass of the GL decoder alone (dxt_decoder.c:211-240,466-515, rgba_to_yuv422.glsl): in = w x h RGBA8, out = UYVY */
static int run_rgba2uyvy(const char *dir, int w, int h, const char *in_path, const char *out_path)
{
        const size_t in_len = (size_t) w * h * 4;
        uint8_t *in = (uint8_t *) malloc(in_len);
        FILE *f = fopen(in_path, "rb");
        if (!f || fread(in, 1, in_len, f) != in_len) { fprintf(stderr, "cannot read %zu bytes from %s\n", in_len, in_path); return 1; }
        fclose(f);
        if (make_context()) return 2;
        GLuint itex, otex, fbo;
        p_glGenTextures(1, &itex);
        p_glBindTexture(GL_TEXTURE_2D, itex);
        tex_params();
        p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, w, h, 0, GL_RGBA, GL_UNSIGNED_INT_8_8_8_8_REV, in);
        p_glGenTextures(1, &otex);
        p_glBindTexture(GL_TEXTURE_2D, otex);
        tex_params();
        p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, w / 2, h, 0, GL_RGBA, GL_UNSIGNED_INT_8_8_8_8_REV, 0);
        p_glGenFramebuffers(1, &fbo);
        p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, otex, 0);
        if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fprintf(stderr, "framebuffer incomplete\n"); return 2; }
        const GLuint prog = link_program(compile(GL_VERTEX_SHADER, dir, "compress_vp.glsl", 0, 0), compile(GL_FRAGMENT_SHADER, dir, "rgba_to_yuv422.glsl", 0, 0));
        const GLuint vao = make_vao(prog);
        p_glUseProgram(prog);
        p_glUniform1i(p_glGetUniformLocation(prog, "image"), 0);
        p_glUniform1f(p_glGetUniformLocation(prog, "imageWidth"), (GLfloat) w); /* dxt_decoder.c:243-244 */
        p_glBindTexture(GL_TEXTURE_2D, itex);
        p_glViewport(0, 0, w / 2, h);
        p_glDisable(GL_DEPTH_TEST);
        p_glBindVertexArray(vao);
        p_glDrawArrays(GL_TRIANGLES, 0, 6);
        p_glBindVertexArray(0);
        p_glReadBuffer(GL_COLOR_ATTACHMENT0);
        uint8_t *out = (uint8_t *) calloc(2, (size_t) w * h);
        p_glReadPixels(0, 0, w / 2, h, GL_RGBA, GL_UNSIGNED_INT_8_8_8_8_REV, out);
        p_glFinish();
        if (p_glGetError() != GL_NO_ERROR) { fprintf(stderr, "GL error\n"); return 2; }
        f = fopen(out_path, "wb");
        if (!f || fwrite(out, 2, (size_t) w * h, f) != (size_t) w * h) { perror(out_path); return 1; }
        fclose(f);
        return 0;
}

int main(int argc, char **argv)
{
        if (argc == 2 && !strcmp(argv[1], "probe")) {
                if (make_context()) return 2;
                printf("%s | %s\n", p_glGetString(GL_VERSION), p_glGetString(GL_RENDERER));
                return 0;
        }
        if (argc != 8) {
                fprintf(stderr, "usage: %s <refdir> <dxt5|dxt1|dxt1yuv> <rgb|rgba|yuv444|uyvy> <w> <h> <in.raw> <out.bin>\n       %s probe\n", argv[0], argv[0]);
                return 1;
        }
        const char *dir = argv[1];
        if (!strcmp(argv[2], "rgba2uyvy")) { /* <refdir> rgba2uyvy rgba <w> <h> <in.rgba> <out.uyvy> */
                return run_rgba2uyvy(dir, atoi(argv[4]), atoi(argv[5]), argv[6], argv[7]);
        }
        if (!strncmp(argv[2], "dec", 3)) { /* <refdir> <dec5|dec1|dec1yuv> rgba <w> <h> <in.dxt> <out.rgba> */
                return run_decode(dir, argv[2], atoi(argv[4]), atoi(argv[5]), argv[6], argv[7]);
        }
        const int dxt5 = !strcmp(argv[2], "dxt5"), dxt1yuv = !strcmp(argv[2], "dxt1yuv");
        const char *fmt = argv[3];
        const int w = atoi(argv[4]), h = atoi(argv[5]);
        const int is_rgb = !strcmp(fmt, "rgb"), is_uyvy = !strcmp(fmt, "uyvy"), is_yuv = is_uyvy || !strcmp(fmt, "yuv444");
        const size_t in_len = (size_t) w * h * (is_rgb ? 3 : (is_uyvy ? 2 : 4));
        /* The reference never sets GL_UNPACK_ALIGNMENT: GL reads GL_RGB lines at its default 4-byte row alignment, i.e. from
         * h * ((3 w + 3) & ~3) bytes, whatever the caller packed (a slip of the reference for 3 w % 4 != 0: a skewed picture and a
         * read past its buffer).  The buffer here covers what GL reads (zero-filled); the file may hold either layout -- packed
         * (the reference's behaviour as is) or lines already at GL's stride (oracle/pyoracle.py gl_row_stride). */
        const size_t gl_len = is_rgb ? (size_t) h * (((size_t) 3 * w + 3) & ~(size_t) 3) : in_len;
        uint8_t *in = (uint8_t *) calloc(1, gl_len);
        FILE *f = fopen(argv[6], "rb");
        if (!f || fread(in, 1, gl_len, f) < in_len) { fprintf(stderr, "cannot read %zu bytes from %s\n", in_len, argv[6]); return 1; }
        fclose(f);
        if (make_context()) return 2;

        /* dxt_encoder_create (dxt_encoder.c:235-430) */
        GLuint fbo, fbo_tex, tex;
        p_glGenFramebuffers(1, &fbo);
        p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
        p_glGenTextures(1, &fbo_tex);
        p_glBindTexture(GL_TEXTURE_2D, fbo_tex);
        tex_params();
        p_glTexImage2D(GL_TEXTURE_2D, 0, dxt5 ? GL_RGBA32UI : GL_RGBA16UI, (w + 3) / 4 * 4, (h + 3) / 4, 0, GL_RGBA_INTEGER, GL_INT, 0);
        p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, fbo_tex, 0);
        const GLuint vs = compile(GL_VERTEX_SHADER, dir, "compress_vp.glsl", 0, 0);
        /* DXT1 from YUV input compresses with the YUV->RGB variant (dxt_encoder.c:318-319); DXT1_YUV would use FORMAT_YUV 0 */
        const GLuint fs = compile(GL_FRAGMENT_SHADER, dir, dxt5 ? "compress_dxt5ycocg_fp.glsl" : "compress_dxt1_fp.glsl", 1, is_yuv && !dxt1yuv);
        const GLuint prog = link_program(vs, fs);
        p_glGenTextures(1, &tex);
        p_glBindTexture(GL_TEXTURE_2D, tex);
        tex_params();
        if (is_rgb) p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGB, w, h, 0, GL_RGB, GL_BYTE, NULL);
        else p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, w, h, 0, GL_RGBA, GL_UNSIGNED_BYTE, NULL);
        GLuint prog422 = 0, tex422 = 0, fbo444 = 0, vao422 = 0;
        if (is_uyvy) { /* dxt_prepare_yuv422_shader (dxt_encoder.c:143-233) */
                const GLuint fs422 = compile(GL_FRAGMENT_SHADER, dir, "yuv422_to_yuv444.glsl", 0, 0);
                prog422 = link_program(compile(GL_VERTEX_SHADER, dir, "compress_vp.glsl", 0, 0), fs422);
                vao422 = make_vao(prog422);
                p_glGenTextures(1, &tex422);
                p_glBindTexture(GL_TEXTURE_2D, tex422);
                tex_params();
                p_glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA, w / 2, h, 0, GL_RGBA, GL_UNSIGNED_INT_8_8_8_8_REV, NULL);
                p_glUseProgram(prog422);
                p_glUniform1i(p_glGetUniformLocation(prog422, "image"), 0);
                p_glUniform1f(p_glGetUniformLocation(prog422, "imageWidth"), (GLfloat) w);
                p_glGenFramebuffers(1, &fbo444);
        }
        p_glBindTexture(GL_TEXTURE_2D, tex);
        p_glViewport(0, 0, (w + 3) / 4, h / 4);
        p_glDisable(GL_DEPTH_TEST);
        p_glUseProgram(prog);
        p_glUniform1i(p_glGetUniformLocation(prog, "image"), 0);
        p_glUniform1i(p_glGetUniformLocation(prog, "imageFormat"), is_yuv ? 1 : 0);
        p_glUniform2f(p_glGetUniformLocation(prog, "imageSize"), (GLfloat) w, (GLfloat) ((h + 3) / 4 * 4));
        p_glUniform1f(p_glGetUniformLocation(prog, "textureWidth"), (GLfloat) ((w + 3) / 4 * 4));
        const GLuint vao = make_vao(prog);

        /* dxt_encoder_compress (dxt_encoder.c:450-575) */
        if (is_uyvy) {
                p_glBindFramebuffer(GL_FRAMEBUFFER, fbo444);
                p_glFramebufferTexture2D(GL_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, tex, 0);
                p_glBindTexture(GL_TEXTURE_2D, tex422);
                p_glViewport(0, 0, w, h);
                p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, w / 2, h, GL_RGBA, GL_UNSIGNED_INT_8_8_8_8_REV, in);
                p_glUseProgram(prog422);
                p_glBindVertexArray(vao422);
                p_glDrawArrays(GL_TRIANGLES, 0, 6);
                p_glBindVertexArray(0);
                p_glViewport(0, 0, (w + 3) / 4, h / 4);
                p_glUseProgram(prog);
                p_glBindFramebuffer(GL_FRAMEBUFFER, fbo);
                p_glBindTexture(GL_TEXTURE_2D, tex);
        } else {
                p_glBindTexture(GL_TEXTURE_2D, tex);
                p_glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, w, h, is_rgb ? GL_RGB : GL_RGBA, GL_UNSIGNED_BYTE, in);
        }
        /* dxt_encoder_compress_texture (dxt_encoder.c:598-690) */
        p_glBindTexture(GL_TEXTURE_2D, tex);
        p_glDrawBuffer(GL_COLOR_ATTACHMENT0);
        if (p_glCheckFramebufferStatus(GL_FRAMEBUFFER) != GL_FRAMEBUFFER_COMPLETE) { fprintf(stderr, "framebuffer incomplete\n"); return 2; }
        p_glBindVertexArray(vao);
        p_glDrawArrays(GL_TRIANGLES, 0, 6);
        p_glBindVertexArray(0);
        p_glReadBuffer(GL_COLOR_ATTACHMENT0);
        const int bw = (w + 3) / 4, bh = (h + 3) / 4;
        const size_t out_len = (size_t) bw * bh * (dxt5 ? 16 : 8);
        uint8_t *out = (uint8_t *) calloc(1, out_len);
        p_glReadPixels(0, 0, bw, bh, GL_RGBA_INTEGER, dxt5 ? GL_UNSIGNED_INT : GL_UNSIGNED_SHORT, out);
        p_glFinish();
        const GLenum e = p_glGetError();
        if (e != GL_NO_ERROR) { fprintf(stderr, "GL error 0x%x\n", e); return 2; }
        f = fopen(argv[7], "wb");
        if (!f || fwrite(out, 1, out_len, f) != out_len) { perror(argv[7]); return 1; }
        fclose(f);
        return 0;
}
