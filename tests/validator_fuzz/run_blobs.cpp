// TEST-ONLY: runs program blobs that hnb_program_validate ACCEPTED through the host build of the product
// interpreters with every register access bounds-checked (HNB_VM_BOUNDS_CHECK) and under AddressSanitizer
// (heap-allocated parameter block, attribute table and slab of the exact sizes the header declares).
// An accepted blob must not be able to touch anything outside its files: any report here is a validator bug.
//   run_blobs <dir>     runs every *.blob of <dir>; prints the file name first, so a crash names its input
#include <dirent.h>
#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include "../cpu_vm/cpu_vm.cpp"

extern "C" void hnb_vm_oob(uint32_t index, uint32_t size) {
    fprintf(stderr, "REGISTER FILE ACCESS OUT OF BOUNDS: index %u, file size %u\n", index, size);
    abort();
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    DIR* d = opendir(argv[1]);
    if (!d) return 2;
    int n = 0;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() < 5 || name.substr(name.size() - 5) != ".blob") continue;
        std::ifstream f(std::string(argv[1]) + "/" + name, std::ios::binary);
        std::vector<uint8_t> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        fprintf(stderr, "%s\n", name.c_str());
        CpuVm* v = cvm_create(blob.data(), blob.size(), 7);
        if (!v) { fprintf(stderr, "cpu_vm rejected an accepted blob\n"); return 3; }
        if (v->h.flags & HNB_PROG_READS_PARENT) { cvm_destroy(v); ++n; continue; }  // needs a parent effect: hnb_simulate refuses to run it without one
        const float sim[6] = {0.5f, 1.0f / 60.0f, 0.5f, 1.0f / 60.0f, 0.5f, 1.0f / 60.0f};
        for (int frame = 0; frame < 3; ++frame)
            for (int generic = 0; generic < 2; ++generic) cvm_step(v, sim, v->h.capacity / 2 + 1, 0x1234u + frame, nullptr, generic);
        cvm_destroy(v);
        ++n;
    }
    closedir(d);
    printf("%d blobs ran clean\n", n);
    return 0;
}
