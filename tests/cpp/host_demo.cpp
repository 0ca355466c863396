// host_demo.cpp -- a compiled host above the C ABI: builds scenes/rtweekend1.ssml's content through
// rt_hip.hpp, renders it with HipSampler::sample_image + the running-mean callback, writes the mean
// image as raw f32 (and a PNG through the output stage).  Driven by tests/test_gpu_parity.py.
//   host_demo <out.f32> <out.png> <width> <height> <spp> <batch> [device,device,...]
// With a device list the Bvh is replicated over those GPUs (rt_scene_create_multi) and every batch is rendered by all of them.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <vector>
#include <memory>

#include "rt_hip.hpp"

int main(int argc, char **argv)
{
	if (argc < 7) {
		std::fprintf(stderr, "usage: host_demo out.f32 out.png width height spp batch\n");
		return 2;
	}
	using namespace rt_hip;
	try {
		// scenes/rtweekend1.ssml:1-43
		SceneBuilder scene;
		const uint32_t sky_tex = scene.lerp({0.5f, 0.7f, 1.0f}, {1.0f, 1.0f, 1.0f});
		const uint32_t grey = scene.solid({0.5f, 0.5f, 0.5f});
		const uint32_t ground = scene.lambertian(grey, 1.0f);
		scene.sphere({0.0f, 1.0f, -100.5f}, 100.0f, ground);
		scene.sphere({0.0f, 1.0f, 0.0f}, 0.5f, ground);
		scene.sky(sky_tex, 100, 100);
		std::vector<int> devices;
		if (argc > 7)
			for (const char *p = argv[7]; *p;) {
				devices.push_back((int)std::strtol(p, const_cast<char **>(&p), 10));
				if (*p == ',')
					++p;
			}
		std::unique_ptr<Bvh> owned(devices.empty() ? new Bvh(scene, SplitType::Sah, 0) : new Bvh(scene, devices));
		Bvh &bvh = *owned;
		SimpleCamera camera({0, 0, 0}, {0, 1, 0}, {0, 0, 1}, 121.28449291441745f, 16.0f / 9.0f, 0.0f, 1.0f);

		RenderOptions o;
		o.width = std::strtoull(argv[3], nullptr, 10);
		o.height = std::strtoull(argv[4], nullptr, 10);
		o.samples_per_pixel = std::strtoull(argv[5], nullptr, 10);
		HipSampler sampler;
		sampler.batch = std::strtoull(argv[6], nullptr, 10);

		Presentation image(o.width * o.height);
		int calls = 0;
		sampler.sample_image(o, camera, bvh, &image, [&](Presentation *p, const SamplerProgressRef &prev, uint64_t i) {
			++calls;
			return running_mean(p, prev, i);
		});
		std::printf("devices %u ", bvh.device_count());
		std::printf("nodes %llu lights %zu calls %d samples %llu rays %llu\n", (unsigned long long)bvh.number_nodes(), bvh.lights().size(),
		            calls, (unsigned long long)image.sampler_progress.samples_completed, (unsigned long long)image.sampler_progress.rays_shot);
		FILE *f = std::fopen(argv[1], "wb");
		std::fwrite(image.sampler_progress.current_image.data(), sizeof(float), image.sampler_progress.current_image.size(), f);
		std::fclose(f);
		check(rt_output_save(argv[2], image.sampler_progress.current_image.data(), (uint32_t)o.width, (uint32_t)o.height, o.gamma));
	} catch (const Error &e) {
		std::fprintf(stderr, "%s\n", e.what());
		return 1;
	}
	return 0;
}
