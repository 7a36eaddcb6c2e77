// zkgpu_jitc SRC OUT -- compiles one run-time generated constraint kernel with hipRTC in a process of its own.
//
// hipRTC (comgr) compiles one program at a time per process, whatever the number of host threads; a setup has three to five step
// programs of seconds each (csrc/expr_jit.hip, csrc/stark_prover.hip setup_new).  libzkgpu therefore starts one of these per program
// when it compiles a setup's programs side by side: same compiler, same options, same code object as the in-process path, which stays
// the fallback when this helper is missing.  Exit status: 0 = the code object is in OUT; 1 = hipRTC REJECTED THE TEXT (its log on stderr);
// 2 = usage; 3 = anything else (SRC unreadable, OUT unwritable -- a full or read-only $TMPDIR --, hiprtcCreateProgram failed): the caller
// then compiles in process instead of reporting a compile error that never happened.
#include <hip/hiprtc.h>
#include <cstdio>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: zkgpu_jitc SRC OUT [option ...]\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 3; }
    std::string src; char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) src.append(buf, n);
    fclose(f);
    std::vector<const char*> opts(argv + 3, argv + argc);
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "zk_eval.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { fprintf(stderr, "hiprtcCreateProgram failed\n"); return 3; }
    if (hiprtcCompileProgram(prog, (int)opts.size(), opts.data()) != HIPRTC_SUCCESS) {
        size_t ln = 0; hiprtcGetProgramLogSize(prog, &ln);
        std::string log(ln, '\0'); if (ln) hiprtcGetProgramLog(prog, &log[0]);
        fprintf(stderr, "%s\n", log.c_str());
        return 1;
    }
    size_t sz = 0; hiprtcGetCodeSize(prog, &sz);
    std::vector<char> code(sz);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 3; }
    const bool ok = fwrite(code.data(), 1, code.size(), o) == code.size();
    return (fclose(o) == 0 && ok) ? 0 : 3;
}
