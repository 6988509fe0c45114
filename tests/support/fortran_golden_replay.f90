! Golden replay through the Fortran boundary (GPU required): every assertion of the reference's own test programs that goes
! through the public module -- the `in`, `match`, `regex`, `validate` and `error` records of tests/golden/ref_tests.tsv, recorded
! from the REAL reference -- pushed through `use forgex` of the drop-in module and compared with the recorded outputs
! (flags, substring, length, from, to, status, error message).  The `prefix` / `suffix` records exercise extract_literal, which the
! module does not export (reference src/forgex.F90:24-28); they are replayed through the C ABI by the Python tests.
!
!     fortran_golden_replay <path to ref_tests.tsv>
program fortran_golden_replay
   use :: forgex
   implicit none
   character(:), allocatable :: path, line, kind
   character(:), allocatable :: f(:)
   character(65536) :: buf
   integer :: u, ios, nf, n_in, n_match, n_regex, n_validate, n_error, n_bad, n_skipped, arglen
   logical :: pure_ok

   call get_command_argument(1, length=arglen)
   if (arglen <= 0) then
      print '(a)', 'usage: fortran_golden_replay <ref_tests.tsv>'
      error stop 2
   end if
   allocate(character(arglen) :: path)
   call get_command_argument(1, path)
   open(newunit=u, file=path, status='old', action='read', iostat=ios)
   if (ios /= 0) then
      print '(a)', 'cannot open '//path
      error stop 2
   end if
   n_in = 0; n_match = 0; n_regex = 0; n_validate = 0; n_error = 0; n_bad = 0; n_skipped = 0
   do
      read(u, '(a)', iostat=ios) buf
      if (ios /= 0) exit
      line = trim(buf)
      if (len(line) == 0) cycle
      if (line(1:1) == '#') cycle
      call split_tabs(line, f, nf)
      if (nf < 2) cycle
      kind = trim(f(2))
      select case (kind)
      case ('match')
         call do_match(unhex(trim(f(3))), unhex(trim(f(4))), trim(f(6)))
      case ('in')
         call do_in(unhex(trim(f(3))), unhex(trim(f(4))), trim(f(6)), to_int(f(7)), to_int(f(8)), to_int(f(9)), to_int(f(10)))
      case ('regex')
         call do_regex(unhex(trim(f(3))), unhex(trim(f(4))), unhex(trim(f(6))), to_int(f(7)), to_int(f(8)), to_int(f(9)), to_int(f(10)))
      case ('validate')
         call do_validate(unhex(trim(f(3))), trim(f(5)))
      case ('error')
         call do_error(unhex(trim(f(3))), to_int(f(5)), unhex(trim(f(6))))
      case default
         n_skipped = n_skipped + 1
      end select
   end do
   close(u)
   pure_ok = callable_from_pure('[a-z]+\d+', 'ab12  cd345')
   if (.not. pure_ok) n_bad = n_bad + 1
   print '(a,i0,a,i0,a,i0,a,i0,a,i0,a,i0,a,i0)', 'replayed: in ', n_in, ', match ', n_match, ', regex ', n_regex, ', validate ', &
      n_validate, ', error ', n_error, '; not through the module (prefix/suffix): ', n_skipped, '; mismatches ', n_bad
   if (n_bad == 0 .and. n_in + n_match + n_regex + n_validate + n_error > 1300) then
      print '(a)', 'FORTRAN GOLDEN REPLAY OK'
   else
      print '(a)', 'FORTRAN GOLDEN REPLAY FAILED'
      error stop 1
   end if

contains

   !> the public names keep the reference's purity: this is a PURE function that uses the operators, regex_f and is_valid_regex
   pure function callable_from_pure(pattern, text) result(ok)
      character(*), intent(in) :: pattern, text
      logical :: ok
      character(:), allocatable :: sub
      integer :: i
      logical :: hits(3)
      ok = pattern .in. text
      ok = ok .and. .not. (pattern .match. text)
      sub = regex_f(pattern, text)
      ok = ok .and. sub == 'ab12'
      ok = ok .and. is_valid_regex(pattern)
      do concurrent (i = 1:3)
         hits(i) = pattern .in. text(i:)
      end do
      ok = ok .and. all(hits)
   end function callable_from_pure

   subroutine report(what, pat, txt)
      character(*), intent(in) :: what, pat, txt
      n_bad = n_bad + 1
      if (n_bad <= 20) print '(a)', 'MISMATCH '//what//' pattern=['//pat//'] text=['//txt//']'
   end subroutine report

   subroutine do_match(pat, txt, got)
      character(*), intent(in) :: pat, txt, got
      n_match = n_match + 1
      if ((pat .match. txt) .neqv. (got == 'T')) call report('match', pat, txt)
   end subroutine do_match

   subroutine do_in(pat, txt, got, from0, to0, length0, status0)
      character(*), intent(in) :: pat, txt, got
      integer, intent(in) :: from0, to0, length0, status0
      character(:), allocatable :: res
      integer :: length, from, to, status
      n_in = n_in + 1
      if ((pat .in. txt) .neqv. (got == 'T')) call report('in', pat, txt)
      length = -1; from = -1; to = -1; status = -1
      call regex(pat, txt, res, length=length, from=from, to=to, status=status)
      if (from /= from0 .or. to /= to0 .or. length /= length0 .or. status /= status0) call report('in/regex span', pat, txt)
   end subroutine do_in

   subroutine do_regex(pat, txt, got, from0, to0, length0, status0)
      character(*), intent(in) :: pat, txt, got
      integer, intent(in) :: from0, to0, length0, status0
      character(:), allocatable :: res
      integer :: length, from, to, status
      n_regex = n_regex + 1
      length = -1; from = -1; to = -1; status = -1
      call regex(pat, txt, res, length=length, from=from, to=to, status=status)
      if (len(res) /= len(got)) then
         call report('regex substring length', pat, txt)
      else if (res /= got) then
         call report('regex substring', pat, txt)
      end if
      if (from /= from0 .or. to /= to0 .or. length /= length0 .or. status /= status0) call report('regex span', pat, txt)
      if (regex_f(pat, txt) /= got) call report('regex_f', pat, txt)
   end subroutine do_regex

   subroutine do_validate(pat, got)
      character(*), intent(in) :: pat, got
      n_validate = n_validate + 1
      if (is_valid_regex(pat) .neqv. (got == 'T')) call report('validate', pat, '')
   end subroutine do_validate

   subroutine do_error(pat, code0, msg0)
      character(*), intent(in) :: pat, msg0
      integer, intent(in) :: code0
      character(:), allocatable :: res
      character(256) :: msg
      integer :: status
      n_error = n_error + 1
      status = -1
      msg = ''
      call regex(pat, '', res, status=status, err_msg=msg)
      if (status /= code0) call report('error code', pat, '')
      if (trim(msg) /= msg0) call report('error message', pat, trim(msg))
   end subroutine do_error

   subroutine split_tabs(s, fields, nfields)
      character(*), intent(in) :: s
      character(:), allocatable, intent(out) :: fields(:)
      integer, intent(out) :: nfields
      integer :: i, start, k, longest
      nfields = 1
      longest = 0
      start = 1
      do i = 1, len(s)
         if (s(i:i) == achar(9)) then
            nfields = nfields + 1
            longest = max(longest, i - start)
            start = i + 1
         end if
      end do
      longest = max(longest, len(s) - start + 1, 1)
      allocate(character(longest) :: fields(nfields))
      k = 1
      start = 1
      do i = 1, len(s)
         if (s(i:i) == achar(9)) then
            fields(k) = s(start:i-1)
            k = k + 1
            start = i + 1
         end if
      end do
      fields(k) = s(start:)
   end subroutine split_tabs

   function unhex(h) result(s)
      character(*), intent(in) :: h
      character(:), allocatable :: s
      integer :: i, v
      if (h == '-') then
         s = ''
         return
      end if
      allocate(character(len(h) / 2) :: s)
      do i = 1, len(h) / 2
         read(h(2*i-1:2*i), '(z2)') v
         s(i:i) = char(v)
      end do
   end function unhex

   function to_int(s) result(v)
      character(*), intent(in) :: s
      integer :: v
      read(s, *) v
   end function to_int

end program fortran_golden_replay
