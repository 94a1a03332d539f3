//! Safe wrapper over `unet-offload-sys` shaped after the functions of ikanago/unet that it stands in for:
//!
//! * [`calculate_checksum`] has the signature of `utils::calculate_checksum(data: &[u8], sum: u16) -> u16`
//!   (src/utils.rs:1) and can replace it at its seven call sites;
//! * [`Offload::recv_batch`] decides a whole drained protocol queue at once: what the loop of
//!   `NetProtocol::recv` (src/protocols.rs:83-97) gets from `ipv4::recv` packet by packet;
//! * [`Offload::send_batch`] builds the packets of a batch of `udp::send` / `icmp::send` requests;
//! * [`Pipeline`] keeps two batches in flight from the stack's single main thread.
//!
//! NOT compiled where it was written (no rustc in that build image): treat it as reviewed source, not as a
//! tested crate.  The C ABI underneath is tested (ctypes and C++ callers with the same argument order and types).

use std::ffi::CStr;
use std::sync::{Mutex, OnceLock};

use anyhow::{bail, ensure, Result};
use unet_offload_sys as sys;

pub use sys::{UnetIface as Iface, UnetResult as FrameResult, UnetTxDesc as TxDesc};

/// `enum unet_verdict`: the first failing check of the reference's receive path, in its order of evaluation.
#[repr(u8)]
#[derive(Clone, Copy, Debug, PartialEq, Eq, Hash)]
pub enum Verdict {
    None = 0,
    DropL2Truncated = 1,
    DropL2Ethertype = 2,
    DropL2NotForMe = 3,
    PassArp = 4,
    DropL3Truncated = 5,
    DropL3Proto = 6,
    DropL3Version = 7,
    DropL3IhlShort = 8,
    DropL3IhlLong = 9,
    DropL3Fragment = 10,
    DropL3Csum = 11,
    IgnoreL3NotForMe = 12,
    DropIcmpTruncated = 13,
    DropIcmpType = 14,
    AcceptIcmpEcho = 15,
    AcceptIcmpEchoReply = 16,
    DropUdpShort = 17,
    DropUdpLen = 18,
    DropUdpCsum = 19,
    AcceptUdp = 20,
    DropTcpShort = 21,
    DropTcpDoff = 22,
    DropTcpCsum = 23,
    AcceptTcp = 24,
    L3HeaderOk = 25,
    DropArpTruncated = 26,
    DropArpHtype = 27,
    DropArpPtype = 28,
    DropArpHlen = 29,
    DropArpPlen = 30,
    IgnoreArpNotForMe = 31,
    DropTcpOption = 32,
}

impl Verdict {
    pub fn from_code(code: u8) -> Option<Verdict> {
        if code <= Verdict::DropTcpOption as u8 {
            // SAFETY: the enum is repr(u8) and has a variant for every value from 0 to DropTcpOption.
            Some(unsafe { std::mem::transmute::<u8, Verdict>(code) })
        } else {
            None
        }
    }

    /// The message the reference attaches to the error it returns for this outcome (`None` where it returns
    /// `Ok(())` or panics).
    pub fn reference_error(self) -> Option<&'static str> {
        match self {
            Verdict::DropL2Ethertype => Some("unknown network protocol type"), // src/protocols.rs:37-40
            Verdict::DropL2NotForMe => Some("ethernet frame not for me"),      // src/devices/ethernet.rs:63-71
            Verdict::DropL3Proto => Some("unknown transport protocol number"), // src/transport.rs:22-25
            Verdict::DropL3Version => Some("invalid version"),                 // src/protocols/ipv4.rs:129-133
            Verdict::DropL3IhlShort => Some("header too short"),               // :134-138
            Verdict::DropL3IhlLong => Some("header too long"),                 // :139-143
            Verdict::DropL3Fragment => Some("fragmentation is not supported"), // :145-147
            Verdict::DropL3Csum => Some("invalid checksum"),                   // :152-157
            Verdict::DropIcmpType => Some("unknown icmp type"),                // src/transport/icmp.rs:16-22
            Verdict::DropUdpShort => Some("udp packet too short"),             // src/transport/udp.rs:226-228
            Verdict::DropUdpLen => Some("invalid udp packet length"),          // :232-238
            Verdict::DropUdpCsum => Some("invalid udp checksum"),              // :247-254
            _ => None,
        }
    }
}

fn last_error() -> String {
    // SAFETY: unet_last_error returns a pointer to a NUL-terminated thread-local buffer that stays valid.
    unsafe { CStr::from_ptr(sys::unet_last_error()) }.to_string_lossy().into_owned()
}

fn check(rc: i32, what: &str) -> Result<()> {
    ensure!(rc == 0, "{what}: {} (code {rc})", last_error());
    Ok(())
}

/// Records of one batch, each starting on a 16-byte boundary of a 16-byte aligned buffer (what every batch entry
/// point of the library takes).  `Vec<u8>` gives no alignment promise, so the bytes live in a `Vec<u128>`.
#[derive(Default)]
pub struct Arena {
    chunks: Vec<u128>,
    off16: Vec<u32>,
    len: Vec<u32>,
}

impl Arena {
    pub fn new() -> Self {
        Self::default()
    }

    pub fn clear(&mut self) {
        self.chunks.clear();
        self.off16.clear();
        self.len.clear();
    }

    pub fn records(&self) -> usize {
        self.len.len()
    }

    /// Appends one record (an L3 packet or an Ethernet frame of at most 65535 bytes).
    pub fn push(&mut self, record: &[u8]) -> Result<()> {
        ensure!(record.len() <= 65535, "record of {} bytes: the limit is 65535", record.len());
        ensure!(self.chunks.len() < u32::MAX as usize, "arena full");
        self.off16.push(self.chunks.len() as u32);
        self.len.push(record.len() as u32);
        for piece in record.chunks(16) {
            let mut c = [0u8; 16];
            c[..piece.len()].copy_from_slice(piece);
            self.chunks.push(u128::from_le_bytes(c));
        }
        Ok(())
    }

    /// Record `i` as the caller pushed it (or as `send_batch` built it).
    pub fn record(&self, i: usize) -> &[u8] {
        let start = self.off16[i] as usize * 16;
        // SAFETY: u128 has no padding and any bit pattern is a valid u8; the range lies inside the vector because
        // push / unet_tx_build place every record, with its pad to 16 bytes, inside it.
        let all = unsafe { std::slice::from_raw_parts(self.chunks.as_ptr() as *const u8, self.chunks.len() * 16) };
        &all[start..start + self.len[i] as usize]
    }

    fn bytes(&self) -> (*const u8, u64) {
        (self.chunks.as_ptr() as *const u8, (self.chunks.len() * 16) as u64)
    }
}

/// One context on one GPU.  Like the reference's packet path it is meant to be driven by one thread.
pub struct Offload {
    ctx: *mut sys::UnetCtx,
    spare: Vec<u128>,
}

// The context may move between threads; it must not be used by two at once (no `Sync`).
unsafe impl Send for Offload {}

impl Offload {
    pub fn open(device: i32) -> Result<Self> {
        let mut ctx = std::ptr::null_mut();
        // SAFETY: out pointer is valid for one write.
        check(unsafe { sys::unet_open(device, &mut ctx) }, "unet_open")?;
        Ok(Offload { ctx, spare: Vec::new() })
    }

    /// `utils::calculate_checksum(data, sum)` (src/utils.rs:1-17) computed by the GPU.  One buffer per call costs
    /// a host-device round trip: keep the CPU function for single headers and batch with `csum_batch` instead.
    pub fn calculate_checksum(&mut self, data: &[u8], sum: u16) -> Result<u16> {
        ensure!(data.len() <= 65535, "buffer of {} bytes: the limit is 65535", data.len());
        let mut out = 0u16;
        // SAFETY: data is valid for len bytes, out for one u16.
        check(unsafe { sys::unet_csum(self.ctx, data.as_ptr(), data.len(), sum, &mut out) }, "unet_csum")?;
        Ok(out)
    }

    /// `calculate_checksum(record, seed)` for every record of an arena (seeds: none = all zero).
    pub fn csum_batch(&mut self, arena: &Arena, seeds: Option<&[u16]>) -> Result<Vec<u16>> {
        let n = arena.records();
        if let Some(s) = seeds {
            ensure!(s.len() == n, "{} seeds for {} records", s.len(), n);
        }
        let mut out = vec![0u16; n];
        let (p, bytes) = arena.bytes();
        let seeds_ptr = seeds.map_or(std::ptr::null(), |s| s.as_ptr());
        // SAFETY: tables hold n entries, the arena holds every record with its pad to 16 bytes.
        check(
            unsafe {
                sys::unet_csum_batch(self.ctx, p, bytes, arena.off16.as_ptr(), 0, arena.len.as_ptr(), seeds_ptr, n as u32, out.as_mut_ptr())
            },
            "unet_csum_batch",
        )?;
        Ok(out)
    }

    /// What `ethernet::recv` (L2 mode) / `ipv4::recv` and then `udp::recv` / `icmp::recv` decide for each record.
    pub fn verify(&mut self, arena: &Arena, iface: &Iface) -> Result<Vec<FrameResult>> {
        let n = arena.records();
        let mut out = vec![FrameResult::default(); n];
        if n == 0 {
            return Ok(out);
        }
        let (p, bytes) = arena.bytes();
        // SAFETY: as in csum_batch; out holds n results.
        check(
            unsafe { sys::unet_verify(self.ctx, p, bytes, arena.off16.as_ptr(), 0, arena.len.as_ptr(), n as u32, iface, out.as_mut_ptr()) },
            "unet_verify",
        )?;
        Ok(out)
    }

    /// The same through the library's static device buffer: no allocation on the way (at most 65,536 records
    /// inside 8 MiB per call).
    pub fn verify_static(&mut self, arena: &Arena, iface: &Iface) -> Result<Vec<FrameResult>> {
        let n = arena.records();
        let mut out = vec![FrameResult::default(); n];
        if n == 0 {
            return Ok(out);
        }
        let (p, bytes) = arena.bytes();
        // SAFETY: as in verify.
        check(
            unsafe { sys::unet_verify_static(self.ctx, p, bytes, arena.off16.as_ptr(), 0, arena.len.as_ptr(), n as u32, iface, out.as_mut_ptr()) },
            "unet_verify_static",
        )?;
        Ok(out)
    }

    /// Drop-in for the body of `NetProtocol::recv`'s drain loop (src/protocols.rs:83-97): the caller drains its
    /// `VecDeque<Ipv4QueueEntry>` into a slice of packet buffers that arrived on one interface and gets one
    /// result per packet, in order.  The host-side follow-ups stay with the caller: PCB lookup and enqueue for
    /// `AcceptUdp` (src/transport/udp.rs:263-277), the echo reply for `AcceptIcmpEcho`
    /// (src/transport/icmp.rs:110-120; `result.csum` already is the reply's ICMP checksum), logging for drops.
    pub fn recv_batch<B: AsRef<[u8]>>(&mut self, packets: &[B], unicast: u32, broadcast: u32) -> Result<Vec<FrameResult>> {
        let mut arena = Arena { chunks: std::mem::take(&mut self.spare), off16: Vec::new(), len: Vec::new() };
        arena.clear();
        for p in packets {
            arena.push(p.as_ref())?;
        }
        let iface = Iface { unicast, broadcast, mac: [0; 6], mode: sys::UNET_MODE_L3 as u8, ext_tcp: 0 };
        let res = self.verify(&arena, &iface);
        self.spare = arena.chunks;
        res
    }

    /// Fills the IPv4 header checksum and the UDP / ICMP checksum of every packet in place
    /// (src/protocols/ipv4.rs:366-391, src/transport/udp.rs:181-203, src/transport/icmp.rs:75-85).
    /// Returns one status per packet: 0 = filled, else the verdict code of the reason it was left alone.
    pub fn generate(&mut self, arena: &mut Arena) -> Result<Vec<u8>> {
        let n = arena.records();
        let mut status = vec![0u8; n];
        if n == 0 {
            return Ok(status);
        }
        let bytes = (arena.chunks.len() * 16) as u64;
        // SAFETY: the arena is exclusively borrowed and large enough for every record.
        check(
            unsafe {
                sys::unet_generate(self.ctx, arena.chunks.as_mut_ptr() as *mut u8, bytes, arena.off16.as_ptr(), 0, arena.len.as_ptr(), n as u32, status.as_mut_ptr())
            },
            "unet_generate",
        )?;
        Ok(status)
    }

    /// Whole packets (L3 mode) or Ethernet frames (L2 mode) from payloads and descriptors: what `udp::send` /
    /// `icmp::send` -> `ipv4::send` -> `tap::send` build one at a time (src/transport/udp.rs:168-216,
    /// src/transport/icmp.rs:66-95, src/protocols/ipv4.rs:300-391, src/driver/tap.rs:132-166).  The decisions those
    /// functions take on the host stay with the caller and arrive in the descriptor: route and source-address
    /// checks, the identification counter, the resolved destination MAC.  Returns the packets as an arena (record
    /// i is packet i; length 0 where the status is not 0) and one `UNET_TX_*` status per packet.
    pub fn send_batch(&mut self, payloads: &Arena, desc: &[TxDesc], iface: &Iface) -> Result<(Arena, Vec<u8>)> {
        let n = payloads.records();
        ensure!(desc.len() == n, "{} descriptors for {} payloads", desc.len(), n);
        let mut out = Arena::new();
        let mut status = vec![0u8; n];
        if n == 0 {
            return Ok((out, status));
        }
        let mut out_bytes = 0u64;
        out.off16 = vec![0u32; n];
        out.len = vec![0u32; n];
        // SAFETY: the tables hold n entries.
        check(
            unsafe { sys::unet_tx_layout(payloads.len.as_ptr(), n as u32, iface.mode as i32, out.off16.as_mut_ptr(), &mut out_bytes) },
            "unet_tx_layout",
        )?;
        out.chunks = vec![0u128; (out_bytes / 16) as usize];
        let (p, bytes) = payloads.bytes();
        // SAFETY: the output arena has the size unet_tx_layout asked for; every table holds n entries.
        check(
            unsafe {
                sys::unet_tx_build(
                    self.ctx, p, bytes, payloads.off16.as_ptr(), 0, payloads.len.as_ptr(), desc.as_ptr(), n as u32, iface,
                    out.chunks.as_mut_ptr() as *mut u8, out_bytes, out.off16.as_mut_ptr(), out.len.as_mut_ptr(), status.as_mut_ptr(),
                )
            },
            "unet_tx_build",
        )?;
        Ok((out, status))
    }

    /// Per-verdict counts of everything this context has judged through host-pointer calls.
    pub fn counters(&mut self, reset: bool) -> Result<[u64; sys::UNET_V_COUNT as usize]> {
        let mut out = [0u64; sys::UNET_V_COUNT as usize];
        // SAFETY: out holds UNET_V_COUNT counters.
        check(unsafe { sys::unet_counters(self.ctx, out.as_mut_ptr(), reset as i32) }, "unet_counters")?;
        Ok(out)
    }
}

impl Drop for Offload {
    fn drop(&mut self) {
        // SAFETY: the context was returned by unet_open and is closed exactly once.
        unsafe { sys::unet_close(self.ctx) }
    }
}

/// Two batches in flight from one thread: `submit` batch k+1 while the GPU works on batch k, `collect` in order.
/// This is how the stack's main thread (src/main.rs:47-57) overlaps packing with GPU work without threads.
pub struct Pipeline<'a> {
    off: &'a mut Offload,
    in_flight: std::collections::VecDeque<(sys::UnetTicket, Box<[FrameResult]>)>,
}

impl<'a> Pipeline<'a> {
    pub fn new(off: &'a mut Offload) -> Self {
        Pipeline { off, in_flight: std::collections::VecDeque::new() }
    }

    /// Hands a batch over; the arena may be reused as soon as this returns.  With two batches already in flight
    /// the oldest is collected first and returned.
    pub fn submit(&mut self, arena: &Arena, iface: &Iface) -> Result<Option<Box<[FrameResult]>>> {
        let mut finished = None;
        if self.in_flight.len() == sys::UNET_ASYNC_DEPTH as usize {
            finished = self.collect()?;
        }
        let n = arena.records();
        let mut out = vec![FrameResult::default(); n].into_boxed_slice();
        let mut ticket = sys::UnetTicket { id: sys::UNET_TICKET_NONE };
        let (p, bytes) = arena.bytes();
        // SAFETY: `out` is heap memory that stays where it is (boxed slice, kept in `in_flight`) until unet_wait
        // for this ticket has returned; the other pointers are only read during the call.
        check(
            unsafe {
                sys::unet_verify_async(self.off.ctx, p, bytes, arena.off16.as_ptr(), 0, arena.len.as_ptr(), n as u32, iface, out.as_mut_ptr(), &mut ticket)
            },
            "unet_verify_async",
        )?;
        self.in_flight.push_back((ticket, out));
        Ok(finished)
    }

    /// Waits for the oldest batch in flight and returns its results (`None` when nothing is in flight).
    pub fn collect(&mut self) -> Result<Option<Box<[FrameResult]>>> {
        match self.in_flight.pop_front() {
            None => Ok(None),
            Some((ticket, out)) => {
                // SAFETY: the ticket came from unet_verify_async on this context and is waited for once.
                check(unsafe { sys::unet_wait(self.off.ctx, ticket) }, "unet_wait")?;
                Ok(Some(out))
            }
        }
    }
}

impl Drop for Pipeline<'_> {
    fn drop(&mut self) {
        // results buffers must outlive the GPU's writes into them
        while let Ok(Some(_)) = self.collect() {}
    }
}

static GLOBAL: OnceLock<Mutex<Offload>> = OnceLock::new();

/// Binds the process-wide context used by [`calculate_checksum`] to a GPU (call once, before the first packet).
pub fn init(device: i32) -> Result<()> {
    let off = Offload::open(device)?;
    if GLOBAL.set(Mutex::new(off)).is_err() {
        bail!("unet_offload::init called twice");
    }
    Ok(())
}

/// Exact signature of `utils::calculate_checksum` (src/utils.rs:1): `use unet_offload::calculate_checksum;` in
/// place of `use crate::utils::calculate_checksum;` at the call sites src/protocols/ipv4.rs:154,387,
/// src/transport/icmp.rs:83 and src/transport/udp.rs:200,202,229,247.  Panics if [`init`] has not run or the GPU
/// call fails, as the reference's function cannot fail either.
pub fn calculate_checksum(data: &[u8], sum: u16) -> u16 {
    let cell = GLOBAL.get().expect("unet_offload::init(device) must run first");
    let mut off = cell.lock().expect("offload context poisoned");
    off.calculate_checksum(data, sum).expect("unet_csum failed")
}

/// `Ipv4Interface::new`'s broadcast rule (src/protocols/ipv4.rs:220).
pub fn broadcast_of(unicast: u32, netmask: u32) -> u32 {
    (unicast & 0xffff_ff00) | (!netmask & 0xff)
}

#[cfg(test)]
mod tests {
    use super::*;

    /// A batch whose only packet is empty: the arena holds no bytes at all, and the library must neither read
    /// past it nor fail.  An empty buffer is shorter than an IPv4 header (the reference indexes out of bounds at
    /// src/protocols/ipv4.rs:189-203; the offload's defined answer is DROP_L3_TRUNCATED).  Needs a GPU.
    #[test]
    #[ignore = "needs an MI355X"]
    fn recv_batch_of_one_empty_packet() {
        let mut off = Offload::open(0).expect("unet_open");
        let res = off.recv_batch(&[&[][..]], 0xC000_0202, 0xC000_02ff).expect("recv_batch");
        assert_eq!(res.len(), 1);
        assert_eq!(res[0].verdict, sys::UNET_V_DROP_L3_TRUNCATED);
    }
}
