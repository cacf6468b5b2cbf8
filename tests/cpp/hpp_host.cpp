// Exercises include/snarkvm_hip.hpp - the C++ host mirror of the reference's Rust plugin crate
// (algorithms/cuda/src/lib.rs:77-168) - exactly as a C++ caller would: built with plain g++ against the C ABI.
//   hpp_host validate            argument checks + host-only corner cases + loud failure without a device (no GPU needed)
//   hpp_host run <dir>           reads bases.bin (104 B stride), scalars.bin (32 B), fr.bin (32 B); writes msm.bin (144 B),
//                                ntt.bin, polymul.bin for the Python test to compare with the oracle
#include <array>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "snarkvm_hip.hpp"

struct Fr { uint64_t l[4]; };
struct G1Affine { uint64_t x[6], y[6]; uint8_t infinity; uint8_t pad[7]; };
struct G1Projective { uint64_t x[6], y[6], z[6]; };
static_assert(sizeof(G1Affine) == 104 && sizeof(G1Projective) == 144 && sizeof(Fr) == 32, "Rust layouts");

template <class T>
static std::vector<T> read_all(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> v(raw.size() / sizeof(T));
    std::memcpy(v.data(), raw.data(), v.size() * sizeof(T));
    return v;
}
template <class T>
static void write_all(const std::string& path, const T* p, size_t n) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)p, (std::streamsize)(n * sizeof(T)));
}

static int validate() {
    int fails = 0;
    std::vector<Fr> x(6);
    try {  // lib.rs:84-86: domain_size must be a power of two
        snarkvm_hip::NTT(6, x.data(), NN, Forward, Standard);
        fails++;
    } catch (const std::invalid_argument&) {
    }
    try {  // lib.rs:150-152: more scalars than points
        std::vector<G1Affine> pts(2);
        std::vector<Fr> sc(3);
        snarkvm_hip::msm<G1Affine, G1Projective, Fr>(pts.data(), pts.size(), sc.data(), sc.size());
        fails++;
    } catch (const std::invalid_argument&) {
    }
    // snarkvm.cu:196-210: a single coefficient-form input is copied through (no device involved)
    std::vector<std::vector<Fr>> polys(1, std::vector<Fr>(3));
    for (int i = 0; i < 3; i++) polys[0][i].l[0] = 7 + i;
    Fr zero{};
    std::vector<Fr> out = snarkvm_hip::polymul<Fr>(4, polys, {}, zero);
    if (out.size() != 4 || out[0].l[0] != 7 || out[2].l[0] != 9 || out[3].l[0] != 0) fails++;
    if (snarkvm_hip_device_count() == 0) {
        try {  // no device: a RustError with a non-zero code, never a silent CPU result
            std::vector<G1Affine> pts(2000);
            std::vector<Fr> sc(2000);
            snarkvm_hip::msm<G1Affine, G1Projective, Fr>(pts.data(), pts.size(), sc.data(), sc.size());
            fails++;
        } catch (const snarkvm_hip::Error& e) {
            if (e.code == 0) fails++;
        }
    }
    std::cout << (fails ? "FAIL " : "OK ") << fails << std::endl;
    return fails;
}

int main(int argc, char** argv) {
    if (argc >= 2 && std::string(argv[1]) == "validate") return validate();
    if (argc >= 3 && std::string(argv[1]) == "run") {
        const std::string dir = argv[2];
        auto bases = read_all<G1Affine>(dir + "/bases.bin");
        auto scalars = read_all<Fr>(dir + "/scalars.bin");
        auto fr = read_all<Fr>(dir + "/fr.bin");
        try {
            G1Projective r = snarkvm_hip::msm<G1Affine, G1Projective, Fr>(bases.data(), bases.size(), scalars.data(), scalars.size());
            write_all(dir + "/msm.bin", &r, 1);
            std::vector<Fr> x = fr;
            snarkvm_hip::NTT(x.size(), x.data(), NN, Forward, Standard);
            write_all(dir + "/ntt.bin", x.data(), x.size());
            std::vector<std::vector<Fr>> polys = {std::vector<Fr>(fr.begin(), fr.begin() + fr.size() / 2),
                                                  std::vector<Fr>(fr.begin() + fr.size() / 2, fr.end())};
            Fr zero{};
            std::vector<Fr> prod = snarkvm_hip::polymul<Fr>(fr.size(), polys, {}, zero);
            write_all(dir + "/polymul.bin", prod.data(), prod.size());
            // the same transform and a commitment over DEVICE-resident operands, every device byte owned through the library (DeviceBuffer:
            // snarkvm_hip_malloc / _memcpy_*), inside a scope: no HIP header, no torch in this process
            snarkvm_hip::DeviceBuffer d_fr(fr.size() * sizeof(Fr)), d_pad(2 * fr.size() * sizeof(Fr)), d_sc(scalars.size() * sizeof(Fr));
            d_fr.upload(fr.data(), d_fr.size());
            d_sc.upload(scalars.data(), d_sc.size());
            snarkvm_hip::RegisteredBases<G1Affine, G1Projective> rb(bases.data(), bases.size(), false, 17, 15);
            G1Projective c{};
            uint32_t lg = 0;
            while (((size_t)1 << lg) < fr.size()) lg++;
            {
                snarkvm_hip::Scope scope(d_fr.data(), SNARKVM_HIP_SCOPE_ASYNC_MSM);
                d_pad.fill(0, 0, d_pad.size());
                d_pad.copy_from(0, d_fr.data(), d_fr.size());
                snarkvm_hip::check(snarkvm_hip_ntt_device(d_pad.data(), lg + 1, NN, Forward, Standard));
                snarkvm_hip::check(snarkvm_hip_msm_registered_ex(&c, rb.handle(), 0, scalars.size(), 0, 0, d_sc.data(), 1, 0, 0));  // enqueued: written by end()
                scope.end();
            }
            std::vector<Fr> xd(2 * fr.size());
            d_pad.download(xd.data(), d_pad.size());
            write_all(dir + "/ntt_dev.bin", xd.data(), xd.size());
            write_all(dir + "/msm_dev.bin", &c, 1);
        } catch (const snarkvm_hip::Error& e) {
            std::cerr << "snarkvm_hip::Error " << e.code << ": " << e.what() << std::endl;
            return 2;
        }
        std::cout << "OK" << std::endl;
        return 0;
    }
    if (argc >= 3 && std::string(argv[1]) == "multi") {
        // Two logical devices (the same GPU listed twice when only one is visible) driven by four concurrent caller threads -
        // the reference's deployment: rayon workers calling the three FFI symbols at once (sonic_pc/mod.rs:203-245), every
        // GPU in use (snarkvm.cu:254-295).  Thread t works on the scalar vector rotated by t and the Fr vector scaled by a
        // rotation, so every result is different; the Python side checks each one against the oracle.
        const std::string dir = argv[2];
        const int ndev_visible = snarkvm_hip_device_count();
        const int32_t ids[2] = {0, ndev_visible > 1 ? 1 : 0};
        try {
            snarkvm_hip::check(snarkvm_hip_set_devices(ids, 2));
            if (snarkvm_hip_num_devices() != 2) throw snarkvm_hip::Error(1, "expected two logical devices");
            auto bases = read_all<G1Affine>(dir + "/bases.bin");
            auto scalars = read_all<Fr>(dir + "/scalars.bin");
            auto fr = read_all<Fr>(dir + "/fr.bin");
            const int T = 4;
            std::vector<std::thread> th;
            std::vector<std::string> errs(T);
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    try {
                        std::vector<Fr> sc(scalars.size());
                        for (size_t i = 0; i < sc.size(); i++) sc[i] = scalars[(i + (size_t)t * 7) % sc.size()];
                        for (int rep = 0; rep < 3; rep++) {  // the repeated base range is registered on both devices (base cache)
                            G1Projective r = snarkvm_hip::msm<G1Affine, G1Projective, Fr>(bases.data(), bases.size(), sc.data(), sc.size());
                            write_all(dir + "/msm_" + std::to_string(t) + "_" + std::to_string(rep) + ".bin", &r, 1);
                        }
                        std::vector<Fr> x(fr.size());
                        for (size_t i = 0; i < x.size(); i++) x[i] = fr[(i + (size_t)t * 3) % x.size()];
                        snarkvm_hip::NTT(x.size(), x.data(), NN, Forward, Standard);
                        write_all(dir + "/ntt_" + std::to_string(t) + ".bin", x.data(), x.size());
                    } catch (const std::exception& e) {
                        errs[t] = e.what();
                    }
                });
            for (auto& x : th) x.join();
            for (auto& e : errs)
                if (!e.empty()) throw snarkvm_hip::Error(1, e);
        } catch (const snarkvm_hip::Error& e) {
            std::cerr << "snarkvm_hip::Error " << e.code << ": " << e.what() << std::endl;
            return 2;
        }
        std::cout << "OK" << std::endl;
        return 0;
    }
    std::cerr << "usage: hpp_host validate | run <dir> | multi <dir>" << std::endl;
    return 64;
}
