! Device-resident batches through `use forgex` (GPU required): type(fx_batch) keeps the rows in HBM across calls and patterns.
!   1. parity: every form that takes a batch against the host-buffer forms of the same module on the same rows
!      (`pattern .in. batch`, `.match.`, `patterns(:) .in. batch`, `call regex(pattern, batch, from, to)`, fx_batch_search + fx_batch_count /
!      fx_batch_fetch), rows of 256 bytes (the headline shape) and of 100 bytes (ragged), invalid pattern;
!   2. rate: fx_batch_search over config-3-like rows resident in HBM (letters / blanks, half of the rows with digits planted in the last
!      quarter; 256-byte rows, count given on the command line, default 4 M), results left on the device, timed with system_clock around
!      `reps` back-to-back calls + one fx_batch_sync -- printed as GB/s of input.
! Written like the reference's own test programs (reference test/test_api/test_case_001.f90): operators, regex.
program fortran_batch_test
   use, intrinsic :: iso_c_binding
   use, intrinsic :: iso_fortran_env, only: int64, real64
   use :: forgex
   implicit none
   integer, parameter :: L = 256
   character(L), allocatable, target :: rows(:)
   character(100), allocatable, target :: rows100(:)
   type(fx_batch) :: batch, b100
   logical, allocatable :: r1(:), r2(:), rm(:, :)
   integer, allocatable :: f1(:), t1(:), f2(:), t2(:)
   integer(c_int8_t), allocatable, target :: flags(:)
   integer(c_int32_t), allocatable, target :: ff(:), tt(:)
   character(16) :: pats(3)
   integer :: n, nsub, i, j, reps, status, narg
   integer(int64) :: c0, c1, rate, seed, nmatch
   real(real64) :: secs, gbs
   logical :: ok
   character(32) :: arg

   n = 4 * 1024 * 1024
   narg = command_argument_count()
   if (narg >= 1) then
      call get_command_argument(1, arg)
      read (arg, *) n
   end if
   nsub = min(n, 100000)
   ok = .true.

   ! ---- rows: a-z (p = 0.9) / blank, half of the rows get 1-3 digits planted at byte 193..253 behind a letter --------------------
   allocate(rows(n))
   seed = 88172645463325252_int64
   do i = 1, n
      do j = 1, L, 8
         call next(seed)
         call fill8(rows(i)(j:j + 7), seed)
      end do
      call next(seed)
      if (iand(seed, 1_int64) == 1_int64) then
         j = 193 + int(iand(ishft(seed, -8), 63_int64)) - 3
         if (j < 193) j = 193
         rows(i)(j - 1:j - 1) = 'q'
         rows(i)(j:j) = achar(48 + int(iand(ishft(seed, -20), 7_int64)))
         if (iand(ishft(seed, -30), 1_int64) == 1_int64) rows(i)(j + 1:j + 1) = '7'
      end if
   end do
   allocate(rows100(nsub))
   do i = 1, nsub
      rows100(i) = rows(i)(150:249)
   end do

   batch = fx_batch_upload(rows)
   b100 = fx_batch_upload(rows100)
   ok = ok .and. fx_batch_size(batch) == n .and. fx_batch_size(b100) == nsub

   ! ---- 1. parity with the host-buffer forms --------------------------------------------------------------------------------------
   r1 = '[a-z]+\d+' .in. batch
   r2 = '[a-z]+\d+' .in. rows(1:nsub)
   if (.not. all(r1(1:nsub) .eqv. r2)) then
      ok = .false.
      print *, '.in. batch differs from .in. rows'
   end if
   nmatch = count(r1)
   r1 = '[a-z ]+\d*[a-z ]*' .match. batch
   r2 = '[a-z ]+\d*[a-z ]*' .match. rows(1:nsub)
   if (.not. all(r1(1:nsub) .eqv. r2)) then
      ok = .false.
      print *, '.match. batch differs'
   end if
   r1 = '[a-z]+\d+' .in. b100
   r2 = '[a-z]+\d+' .in. rows100
   if (.not. all(r1 .eqv. r2)) then
      ok = .false.
      print *, '.in. batch (100-byte rows) differs'
   end if
   allocate(f1(nsub), t1(nsub), f2(nsub), t2(nsub))
   call regex('[a-z]+\d+', b100, f1, t1, status)
   call regex('[a-z]+\d+', rows100, f2, t2)
   if (status /= 0 .or. .not. (all(f1 == f2) .and. all(t1 == t2))) then
      ok = .false.
      print *, 'regex(batch) differs'
   end if
   call regex('a(', b100, f1, t1, status)
   ok = ok .and. status == 2 .and. all(f1 == -9999) .and. all(t1 == -9999)
   r1 = 'a(' .in. b100
   ok = ok .and. .not. any(r1)
   ! an array of patterns against one resident batch: res(i, j) = pats(j) .in. row i
   pats = [character(16) :: '[a-z]+\d+', 'q\d', 'zzz+']
   rm = pats .in. b100
   do j = 1, 3
      r2 = trim(pats(j)) .in. rows100
      if (.not. all(rm(:, j) .eqv. r2)) then
         ok = .false.
         print *, 'patterns(:) .in. batch differs for pattern', j
      end if
   end do
   ! results left on the device: count (a device reduction) and fetch
   call fx_batch_search('[a-z]+\d+', batch, status=status)
   ok = ok .and. status == 0 .and. fx_batch_count(batch) == nmatch
   allocate(flags(n), ff(n), tt(n))
   call fx_batch_fetch(batch, flags, ff, tt)
   call regex('[a-z]+\d+', rows(1:nsub), f2, t2)
   if (.not. (all(int(ff(1:nsub)) == f2) .and. all(int(tt(1:nsub)) == t2) .and. count(flags /= 0) == nmatch)) then
      ok = .false.
      print *, 'fx_batch_fetch differs'
   end if
   call fx_batch_match('[a-z ]+\d*[a-z ]*', batch)
   ok = ok .and. fx_batch_count(batch) == count('[a-z ]+\d*[a-z ]*' .match. batch)

   ! ---- 2. rate over the resident rows -----------------------------------------------------------------------------------------------
   reps = 50
   do i = 1, 10
      call fx_batch_search('[a-z]+\d+', batch)
   end do
   call fx_batch_sync(batch)
   call system_clock(c0, rate)
   do i = 1, reps
      call fx_batch_search('[a-z]+\d+', batch)
   end do
   call fx_batch_sync(batch)
   call system_clock(c1)
   secs = real(c1 - c0, real64) / real(rate, real64)
   gbs = real(n, real64) * real(L, real64) * real(reps, real64) / secs / 1.0e9_real64
   print '(a,i0,a,i0,a,f10.1,a,f8.4,a)', 'RESIDENT RATE rows ', n, ' x ', L, ' B  ', gbs, ' GB/s of input  ', secs / reps * 1.0e3_real64, ' ms per call (flags + spans left on the device)'
   call system_clock(c0)
   do i = 1, reps
      call fx_batch_search('[a-z]+\d+', batch, spans=.false.)
   end do
   call fx_batch_sync(batch)
   call system_clock(c1)
   secs = real(c1 - c0, real64) / real(rate, real64)
   print '(a,f10.1,a)', 'RESIDENT RATE flags only ', real(n, real64) * real(L, real64) * real(reps, real64) / secs / 1.0e9_real64, ' GB/s of input'
   call system_clock(c0)
   r1 = '[a-z]+\d+' .in. batch
   call system_clock(c1)
   secs = real(c1 - c0, real64) / real(rate, real64)
   print '(a,f10.1,a)', 'OPERATOR RATE  pattern .in. batch (verdicts copied back as logical) ', real(n, real64) * real(L, real64) / secs / 1.0e9_real64, ' GB/s of input'
   call system_clock(c0)
   r2 = '[a-z]+\d+' .in. rows(1:nsub)
   call system_clock(c1)
   secs = real(c1 - c0, real64) / real(rate, real64)
   print '(a,f10.1,a)', 'HOST-BUFFER RATE  pattern .in. rows (rows over PCIe every call) ', real(nsub, real64) * real(L, real64) / secs / 1.0e9_real64, ' GB/s of input'

   call fx_batch_free(batch)
   call fx_batch_free(b100)
   if (ok) then
      print '(a)', 'FORTRAN BATCH OK'
   else
      print '(a)', 'FORTRAN BATCH FAILED'
      error stop 1
   end if

contains

   subroutine next(s)   ! xorshift64
      integer(int64), intent(inout) :: s
      s = ieor(s, ishft(s, 13))
      s = ieor(s, ishft(s, -7))
      s = ieor(s, ishft(s, 17))
   end subroutine next

   subroutine fill8(c, s)   ! 8 characters from 64 random bits: a blank with probability 1/8 (close to the generator's 1/10), else a-z
      character(8), intent(out) :: c
      integer(int64), intent(in) :: s
      integer :: k, v
      do k = 1, 8
         v = int(iand(ishft(s, -8 * (k - 1)), 255_int64))
         if (iand(v, 7) == 0) then
            c(k:k) = ' '
         else
            c(k:k) = achar(97 + mod(v / 8, 26))
         end if
      end do
   end subroutine fill8
end program fortran_batch_test
