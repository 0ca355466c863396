// fake_rccl.cpp -- a STAND-IN for librccl with the six entry points librt_hip.so binds (ncclCommInitAll, ncclCommDestroy,
// ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv), for the one-GPU test box: it checks the CALL PATTERN of the multi-device
// gather (csrc/rt_api.cpp render_device_multi) the way the real library's semantics demand -- sends and receives only inside a
// group, every receive matched by a send of the same count and type from the rank it names, buffers that live on the device
// of the communicator they are posted to -- and then moves the bytes with a device copy ordered after the sender's stream, on
// the receiver's stream, as a real ncclRecv would complete there.  Test infrastructure only (tests/test_gpu_parity.py builds
// it with hipcc and points RT_HIP_RCCL_LIB at it); it says nothing about RCCL's own behaviour on xGMI.
//   FAKE_RCCL_REFUSE=1   ncclCommInitAll returns ncclInvalidUsage (5): the "RCCL refuses the device list" path
//   FAKE_RCCL_LOG=file   one line per call / per matched transfer, appended
// Built with -DFAKE_RCCL_NO_RECV the library lacks ncclRecv: the "library lacks a symbol" path.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace {
struct Comm {
	int rank, n, device;
	unsigned magic;
};
struct Op {
	bool send;
	void *buf;
	size_t count;
	int type, peer;
	Comm *comm;
	hipStream_t stream;
};
std::vector<Op> g_ops;
int g_depth = 0;
void log_line(const char *fmt, ...)
{
	const char *path = std::getenv("FAKE_RCCL_LOG");
	if (!path)
		return;
	FILE *f = std::fopen(path, "a");
	if (!f)
		return;
	va_list ap;
	va_start(ap, fmt);
	std::vfprintf(f, fmt, ap);
	va_end(ap);
	std::fputc('\n', f);
	std::fclose(f);
}
bool on_device(const void *p, int device)
{
	hipPointerAttribute_t a;
	if (hipPointerGetAttributes(&a, p) != hipSuccess)
		return false;
	return a.device == device;
}
} // namespace

extern "C" {

int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
	if (std::getenv("FAKE_RCCL_REFUSE")) {
		log_line("init refused n=%d", ndev);
		return 5; // ncclInvalidUsage
	}
	if (!comms || ndev < 1)
		return 4; // ncclInvalidArgument
	for (int r = 0; r < ndev; ++r)
		comms[r] = new Comm{r, ndev, devlist ? devlist[r] : r, 0xC0FFEEu};
	log_line("init n=%d", ndev);
	return 0;
}
int ncclCommDestroy(void *comm)
{
	Comm *c = static_cast<Comm *>(comm);
	if (!c || c->magic != 0xC0FFEEu)
		return 4;
	c->magic = 0;
	log_line("destroy rank=%d", c->rank);
	delete c;
	return 0;
}
int ncclGroupStart()
{
	++g_depth;
	return 0;
}
static int post(bool send, void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream)
{
	Comm *c = static_cast<Comm *>(comm);
	if (g_depth < 1) {
		log_line("ERROR %s outside a group", send ? "send" : "recv");
		return 5;
	}
	if (!c || c->magic != 0xC0FFEEu || peer < 0 || peer >= c->n || peer == c->rank || !buf || count == 0) {
		log_line("ERROR bad %s arguments", send ? "send" : "recv");
		return 4;
	}
	if (!on_device(buf, c->device)) {
		log_line("ERROR %s buffer of rank %d does not live on device %d", send ? "send" : "recv", c->rank, c->device);
		return 4;
	}
	g_ops.push_back(Op{send, buf, count, type, peer, c, stream});
	return 0;
}
int ncclSend(const void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream)
{
	return post(true, const_cast<void *>(buf), count, type, peer, comm, stream);
}
#ifndef FAKE_RCCL_NO_RECV
int ncclRecv(void *buf, size_t count, int type, int peer, void *comm, hipStream_t stream)
{
	return post(false, buf, count, type, peer, comm, stream);
}
#endif
int ncclGroupEnd()
{
	if (g_depth < 1)
		return 5;
	if (--g_depth > 0)
		return 0;
	std::vector<Op> ops;
	ops.swap(g_ops);
	std::vector<bool> used(ops.size(), false);
	int n_pairs = 0;
	for (size_t i = 0; i < ops.size(); ++i) {
		if (ops[i].send)
			continue;
		const Op &r = ops[i];
		size_t j = 0;
		for (; j < ops.size(); ++j)
			if (!used[j] && ops[j].send && ops[j].comm->rank == r.peer && ops[j].peer == r.comm->rank)
				break;
		if (j == ops.size() || ops[j].count != r.count || ops[j].type != r.type || r.type != 7 /* ncclFloat32 */) {
			log_line("ERROR recv at rank %d from %d (%zu elements) has no matching send", r.comm->rank, r.peer, r.count);
			return 3; // ncclInternalError
		}
		used[i] = used[j] = true;
		const Op &s = ops[j];
		// the transfer completes on the receiver's stream, after everything queued before it on the sender's
		hipEvent_t ev;
		if (hipSetDevice(s.comm->device) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
		    hipEventRecord(ev, s.stream) != hipSuccess || hipSetDevice(r.comm->device) != hipSuccess || hipStreamWaitEvent(r.stream, ev, 0) != hipSuccess ||
		    hipMemcpyAsync(r.buf, s.buf, r.count * sizeof(float), hipMemcpyDeviceToDevice, r.stream) != hipSuccess)
			return 1; // ncclUnhandledCudaError
		(void)hipEventDestroy(ev); // (released once the work that uses it has run)
		log_line("transfer rank %d -> rank %d, %zu floats", s.comm->rank, r.comm->rank, r.count);
		++n_pairs;
	}
	for (size_t j = 0; j < ops.size(); ++j)
		if (!used[j]) {
			log_line("ERROR a %s at rank %d was never matched", ops[j].send ? "send" : "recv", ops[j].comm->rank);
			return 3;
		}
	log_line("group: %d transfers", n_pairs);
	return 0;
}

} // extern "C"
