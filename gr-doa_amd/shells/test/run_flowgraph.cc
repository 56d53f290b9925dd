// run_flowgraph — exercises the C++ block shells the way the reference's flowgraphs wire them
// (apps/run_MUSIC_lin_array_simulation.grc, apps/run_RootMUSIC_lin_array_simulation.grc):
//     N stream files -> autocorrelate -> MUSIC_lin_array -> find_local_max   (mode "music")
//     N stream files -> autocorrelate -> rootMUSIC_linear_array               (mode "root")
//     N stream files -> music_pipeline (the three blocks of "music" as one)   (mode "pipeline")
//     N stream files -> root_music_pipeline (the two blocks of "root" as one) (mode "root_pipeline")
// Inputs/outputs are raw little-endian binary files so that the pytest driver (tests/
// test_gpu_shells.py) can compare every port with the Python binding and the oracle.
//
// usage: run_flowgraph music|root <in_prefix> <out_prefix> inputs snapshot overlap avg norm_spacing
//                      num_targets pspectrum_len max_noutput
//   reads  <in_prefix>.ch<k>.c64   (gr_complex samples of stream k, no history)
//   writes <out_prefix>.cov.c64, .spec.f32, .max.f32, .argmax.f32   (music)
//          <out_prefix>.cov.c64, .aoa.f32                            (root)
//          <out_prefix>.spec.f32, .max.f32, .argmax.f32              (pipeline)
//          <out_prefix>.aoa.f32                                      (root_pipeline)
// and prints the wall time spent inside the blocks' work() calls (host buffers in, host buffers out).
#include <doa/MUSIC_lin_array.h>
#include <doa/autocorrelate.h>
#include <doa/find_local_max.h>
#include <doa/music_pipeline.h>
#include <doa/rootMUSIC_linear_array.h>
#include <doa/root_music_pipeline.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

using gr::lite::port_data;

static port_data read_file(const std::string &path, size_t item_size)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    port_data d;
    d.item_size = item_size;
    d.bytes.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return d;
}
static void write_file(const std::string &path, const port_data &d)
{
    std::ofstream f(path, std::ios::binary);
    f.write(d.bytes.data(), (std::streamsize)d.bytes.size());
}

int main(int argc, char **argv)
{
    if (argc != 12) {
        std::cerr << "usage: run_flowgraph music|root|pipeline|root_pipeline in_prefix out_prefix inputs snapshot overlap avg norm_spacing "
                     "num_targets pspectrum_len max_noutput\n";
        return 2;
    }
    const std::string mode = argv[1], in_prefix = argv[2], out_prefix = argv[3];
    const int inputs = atoi(argv[4]), snapshot = atoi(argv[5]), overlap = atoi(argv[6]), avg = atoi(argv[7]);
    const float d = (float)atof(argv[8]);
    const int M = atoi(argv[9]), P = atoi(argv[10]), max_noutput = atoi(argv[11]);
    try {
        std::vector<port_data> streams;
        for (int k = 0; k < inputs; k++) streams.push_back(read_file(in_prefix + ".ch" + std::to_string(k) + ".c64", sizeof(gr_complex)));

        double work_s = 0.0;
        if (mode == "pipeline") {
            auto pipe = gr::doa::music_pipeline::make(inputs, snapshot, overlap, avg, d, M, P);
            auto out = gr::lite::run_block(*pipe, streams, 3, max_noutput);      // first pass: code load, buffer allocation
            write_file(out_prefix + ".argmax.f32", out[0]);
            write_file(out_prefix + ".max.f32", out[1]);
            write_file(out_prefix + ".spec.f32", out[2]);
            auto again = gr::lite::run_block(*pipe, streams, 3, max_noutput, &work_s);   // timed: the same stream once more
            if (again[0].bytes != out[0].bytes || again[2].bytes != out[2].bytes) throw std::runtime_error("music_pipeline: second pass differs");
            std::cout << "items: peaks " << out[0].items() << " spec " << out[2].items() << "\nwork_seconds " << work_s
                      << " snapshots_per_s " << (work_s > 0 ? out[0].items() / work_s : 0.0) << std::endl;
            // the same block with only port 0 connected (angles out: nothing else crosses PCIe on the way back)
            auto pipe1 = gr::doa::music_pipeline::make(inputs, snapshot, overlap, avg, d, M, P);
            double w1 = 0.0;
            auto out1 = gr::lite::run_block(*pipe1, streams, 1, max_noutput);
            out1 = gr::lite::run_block(*pipe1, streams, 1, max_noutput, &w1);
            if (out1[0].bytes != out[0].bytes) throw std::runtime_error("music_pipeline: port 0 differs with ports 1, 2 unconnected");
            std::cout << "angles_only work_seconds " << w1 << " snapshots_per_s " << (w1 > 0 ? out1[0].items() / w1 : 0.0) << std::endl;
            return 0;
        }
        if (mode == "root_pipeline") {
            auto pipe = gr::doa::root_music_pipeline::make(inputs, snapshot, overlap, avg, d, M);
            auto out = gr::lite::run_block(*pipe, streams, 1, max_noutput);      // first pass: code load, buffer allocation
            write_file(out_prefix + ".aoa.f32", out[0]);
            auto again = gr::lite::run_block(*pipe, streams, 1, max_noutput, &work_s);
            if (again[0].bytes != out[0].bytes) throw std::runtime_error("root_music_pipeline: second pass differs");
            std::cout << "items: aoa " << out[0].items() << "\nwork_seconds " << work_s << " snapshots_per_s "
                      << (work_s > 0 ? out[0].items() / work_s : 0.0) << std::endl;
            return 0;
        }
        auto ac = gr::doa::autocorrelate::make(inputs, snapshot, overlap, avg);
        auto cov = gr::lite::run_block(*ac, streams, 1, max_noutput, &work_s);
        write_file(out_prefix + ".cov.c64", cov[0]);

        if (mode == "music") {
            auto music = gr::doa::MUSIC_lin_array::make(d, M, inputs, P);
            auto spec = gr::lite::run_block(*music, {cov[0]}, 1, max_noutput, &work_s);
            write_file(out_prefix + ".spec.f32", spec[0]);
            auto fmax = gr::doa::find_local_max::make(M, P, 0.0f, 180.0f);
            auto pk = gr::lite::run_block(*fmax, {spec[0]}, 2, max_noutput, &work_s);
            write_file(out_prefix + ".max.f32", pk[0]);
            write_file(out_prefix + ".argmax.f32", pk[1]);
            std::cout << "items: cov " << cov[0].items() << " spec " << spec[0].items() << " peaks " << pk[0].items()
                      << "\nwork_seconds " << work_s << " snapshots_per_s " << (work_s > 0 ? pk[0].items() / work_s : 0.0) << std::endl;
        } else {
            auto root = gr::doa::rootMUSIC_linear_array::make(d, M, inputs);
            auto aoa = gr::lite::run_block(*root, {cov[0]}, 1, max_noutput);
            write_file(out_prefix + ".aoa.f32", aoa[0]);
            std::cout << "items: cov " << cov[0].items() << " aoa " << aoa[0].items() << std::endl;
        }
    } catch (const std::exception &e) {
        std::cerr << "run_flowgraph: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
