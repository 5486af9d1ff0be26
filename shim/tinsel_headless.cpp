// tinsel_headless.cpp -- a headless stand-in for the caller of the boundary (reference src/main.cpp,
// which needs GLUT/OpenGL): the reference's OWN loader, Scene::Build and CLI conventions
// (main.cpp:95-172, 174-220, 242-271) driving CreateGpuRenderer() through the reference's
// Renderer interface, with the reference's CreateCpuRenderer() timed beside it.
//
//   tinsel_headless scene.tin [-spp=N] [-width=W] [-height=H] [-maxdepth=D] [-cpuspp=M] [-out=file.pfm]
//                              [-png=file.png] [-nlm=RADIUS]      display stage on the GPU, file by the reference's WritePng
//   tinsel_headless frame%d.tin [-spp=N] ...                     BATCH / animation mode (main.cpp:104-118, 314-327): frames 0, 1, 2, ... until a
//                              file is missing, `<frame file>.png` each (or -png=pattern%d.png).  The reference deletes and re-creates its renderer
//                              per frame; here one renderer lives through the batch and a frame that differs from the last in primitive transforms
//                              only is updated in place (HipRendererUpdateScene, hip_renderer.cpp) -- same PNG bytes as a fresh renderer's.
//
// Prints per-back-end wall time and the image-level difference of the two estimates (different RNG
// streams: statistical agreement only; the per-seed parity tests live in tests/).
#include "render.h"
#include "loader.h"
#include "scene.h"
#include "util.h"
#include "pfm.h"
#include "png.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int HipRendererRenderPasses(Renderer* r, const Camera& camera, const Options& options, Color* output, int passes);
extern "C" int HipRendererPresent(Renderer* r, const Options& options, Color* filtered, int nlmWidth, float nlmFalloff);
extern "C" int HipRendererUpdateScene(Renderer* r, const Scene* prev, const Scene* next);

static void default_options(Options& options, Camera& camera)
{
    // defaults of main.cpp:181-193
    options.width = 512;
    options.height = 256;
    options.filter = Filter(eFilterGaussian, 0.75f, 1.0f);
    options.mode = ePathTrace;
    options.exposure = 1.0f;
    options.limit = 1.5f;
    options.clamp = FLT_MAX;
    options.maxDepth = 4;
    options.maxSamples = INT_MAX;
    camera.position = Vec3(0.0f, 1.0f, 5.0f);
    camera.rotation = Quat();
    camera.fov = DegToRad(35.0f);
}

// main.cpp's batch mode with ONE renderer for the whole animation where the frames allow it
static int batch(int argc, char* argv[], const char* pattern)
{
    int spp = 64, nlmWidth = 0;
    const char* pngPattern = NULL;
    Scene* scenes[2] = { new Scene(), new Scene() };
    Renderer* gpu = NULL;
    int index = 0, inPlace = 0;
    double firstMs = 0.0, inPlaceMs = 0.0;
    for (;; ++index)
    {
        char name[2048], out[2200];
        snprintf(name, sizeof(name), pattern, index);
        Scene* scene = scenes[index & 1];
        Scene* prev = scenes[(index & 1) ^ 1];
        Camera camera;
        Options options;
        default_options(options, camera);
        auto t0 = std::chrono::steady_clock::now();
        FILE* probe = fopen(name, "r");
        if (!probe)
        {
            if (index == 0)
            {
                printf("Couldn't open %s for reading.\n", name);       // (main.cpp:131-135)
                return 1;
            }
            break;
        }
        fclose(probe);
        scene->Clear();
        if (!LoadTin(name, scene, &camera, &options))
            return 1;
        for (int i = 1; i < argc; ++i)
        {
            sscanf(argv[i], "-spp=%d", &spp);
            sscanf(argv[i], "-width=%d", &options.width);
            sscanf(argv[i], "-height=%d", &options.height);
            sscanf(argv[i], "-maxdepth=%d", &options.maxDepth);
            sscanf(argv[i], "-nlm=%d", &nlmWidth);
            if (strncmp(argv[i], "-png=", 5) == 0)
                pngPattern = argv[i] + 5;
        }
        scene->Build();
        const char* how = "created";
        int rc = gpu ? HipRendererUpdateScene(gpu, prev, scene) : 1;
        if (rc < 0)
            return 2;
        if (rc == 0)
            how = "updated in place";
        else
        {
            if (gpu)
            {
                delete gpu;
                how = "re-created (the frame differs in more than transforms)";
            }
            gpu = CreateGpuRenderer(scene);
        }
        gpu->Init(options.width, options.height);
        const double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()*1e3;
        printf("frame %d: %s: renderer %s in %.3fms (loader and Scene::Build included)\n", index, name, how, ms);
        if (index == 0) firstMs = ms;
        if (rc == 0) { ++inPlace; inPlaceMs += ms; }

        const size_t npix = (size_t)options.width*options.height;
        std::vector<Color> pixels(npix), filtered(npix);
        if (HipRendererRenderPasses(gpu, camera, options, &pixels[0], spp) || HipRendererPresent(gpu, options, &filtered[0], nlmWidth, 200.0f))
            return 2;
        if (pngPattern && strchr(pngPattern, '%'))
            snprintf(out, sizeof(out), pngPattern, index);
        else
            snprintf(out, sizeof(out), "%s.png", name);              // (main.cpp:113-115)
        WritePng(&filtered[0], options.width, options.height, out);
        printf("wrote %s\n", out);
        fflush(stdout);
    }
    delete gpu;
    scenes[0]->Clear(); scenes[1]->Clear();
    printf("%d frames; first renderer ready in %.3fms", index, firstMs);
    if (inPlace)
        printf("; %d later frames updated in place in %.3fms on average", inPlace, inPlaceMs/inPlace);
    printf("\n");
    return 0;
}

int main(int argc, char* argv[])
{
    Scene scene;
    Camera camera;
    Options options;

    // defaults of main.cpp:181-193
    options.width = 512;
    options.height = 256;
    options.filter = Filter(eFilterGaussian, 0.75f, 1.0f);
    options.mode = ePathTrace;
    options.exposure = 1.0f;
    options.limit = 1.5f;
    options.clamp = FLT_MAX;
    options.maxDepth = 4;
    options.maxSamples = INT_MAX;
    camera.position = Vec3(0.0f, 1.0f, 5.0f);
    camera.rotation = Quat();
    camera.fov = DegToRad(35.0f);

    int spp = 64, cpuSpp = 0, nlmWidth = 0;
    const char* out = NULL;
    const char* png = NULL;
    const char* file = NULL;

    for (int i = 1; i < argc; ++i)      // "-key=value" overrides after the scene file (main.cpp:143-149)
    {
        if (strstr(argv[i], ".tin") && argv[i][0] != '-')
            file = argv[i];
    }
    if (file && strchr(file, '%'))      // a printf pattern: batch mode (main.cpp:104-118)
        return batch(argc, argv, file);
    if (!file || !LoadTin(file, &scene, &camera, &options))
    {
        printf("usage: tinsel_headless scene.tin [-spp=N] [-width=W] [-height=H] [-maxdepth=D] [-cpuspp=M] [-out=f.pfm]\n");
        return 1;
    }
    for (int i = 1; i < argc; ++i)
    {
        sscanf(argv[i], "-spp=%d", &spp);
        sscanf(argv[i], "-cpuspp=%d", &cpuSpp);
        sscanf(argv[i], "-width=%d", &options.width);
        sscanf(argv[i], "-height=%d", &options.height);
        sscanf(argv[i], "-maxdepth=%d", &options.maxDepth);
        sscanf(argv[i], "-nlm=%d", &nlmWidth);
        if (strncmp(argv[i], "-out=", 5) == 0)
            out = argv[i] + 5;
        if (strncmp(argv[i], "-png=", 5) == 0)
            png = argv[i] + 5;
    }

    scene.Build();      // main.cpp:199

    const size_t npix = (size_t)options.width*options.height;
    std::vector<Color> gpuPixels(npix), cpuPixels(npix);

    Renderer* gpu = CreateGpuRenderer(&scene);
    gpu->Init(options.width, options.height);
    auto t0 = std::chrono::steady_clock::now();
    if (HipRendererRenderPasses(gpu, camera, options, &gpuPixels[0], spp))
    {
        delete gpu;
        return 2;
    }
    auto t1 = std::chrono::steady_clock::now();
    const double gpuSec = std::chrono::duration<double>(t1 - t0).count();
    printf("gpu: %d spp %dx%d in %.4f s (%.2f Msamples/s)\n", spp, options.width, options.height, gpuSec, spp*npix/gpuSec/1e6);
    if (png)
    {
        // main.cpp:258-282 on the device, then the reference's own WritePng (main.cpp:307-312)
        std::vector<Color> filtered(npix);
        if (HipRendererPresent(gpu, options, &filtered[0], nlmWidth, 200.0f) == 0)
        {
            WritePng(&filtered[0], options.width, options.height, png);
            printf("wrote %s\n", png);
        }
    }
    delete gpu;

    if (cpuSpp > 0)
    {
        Renderer* cpu = CreateCpuRenderer(&scene);
        cpu->Init(options.width, options.height);
        auto c0 = std::chrono::steady_clock::now();
        for (int i = 0; i < cpuSpp; ++i)        // main.cpp:246-250
            cpu->Render(camera, options, &cpuPixels[0]);
        auto c1 = std::chrono::steady_clock::now();
        const double cpuSec = std::chrono::duration<double>(c1 - c0).count();
        printf("cpu: %d spp in %.4f s (%.3f Msamples/s, 1 thread)\n", cpuSpp, cpuSec, cpuSpp*npix/cpuSec/1e6);
        delete cpu;

        double sum = 0.0, meanG = 0.0, meanC = 0.0;
        for (size_t i = 0; i < npix; ++i)
        {
            const Color g = gpuPixels[i], c = cpuPixels[i];
            const float gw = g.w > 0.0f ? 1.0f/g.w : 0.0f, cw = c.w > 0.0f ? 1.0f/c.w : 0.0f;     // main.cpp:268
            const float dx = g.x*gw - c.x*cw, dy = g.y*gw - c.y*cw, dz = g.z*gw - c.z*cw;
            sum += dx*dx + dy*dy + dz*dz;
            meanG += (g.x + g.y + g.z)*gw/3.0;
            meanC += (c.x + c.y + c.z)*cw/3.0;
        }
        printf("mean radiance gpu %.5f cpu %.5f ; per-pixel L2 between the two estimates %.4e (independent RNG streams)\n",
               meanG/npix, meanC/npix, std::sqrt(sum/npix));
    }

    if (out)
    {
        PfmImage img;
        img.width = options.width;
        img.height = options.height;
        img.depth = 1;
        std::vector<float> rgb(npix*3);
        for (size_t i = 0; i < npix; ++i)
        {
            const float s = gpuPixels[i].w > 0.0f ? options.exposure/gpuPixels[i].w : 0.0f;
            rgb[i*3 + 0] = gpuPixels[i].x*s; rgb[i*3 + 1] = gpuPixels[i].y*s; rgb[i*3 + 2] = gpuPixels[i].z*s;
        }
        img.data = &rgb[0];
        PfmSave(out, img);
        printf("wrote %s\n", out);
    }
    return 0;
}
