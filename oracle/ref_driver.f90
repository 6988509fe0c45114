! Test infrastructure only -- NOT part of the product.
!
! ref_driver: a small command driver around the REAL reference library (built from
! /root/reference/src by oracle/Makefile into oracle/_ref/libforgex_ref.a).  It calls the
! reference's public API exactly as user code would (`use forgex`: `.in.`, `.match.`,
! `regex`, `is_valid_regex`; reference src/forgex.F90:24-28) plus `extract_literal`
! (reference src/ast/syntax_tree_optimize_m.F90:42-55), and prints the results.
!
! Line protocol on stdin (fields separated by one blank, byte strings hex-encoded, "-" = empty):
!     I <pattern_hex> <text_hex>     ->  I <T|F>
!     M <pattern_hex> <text_hex>     ->  M <T|F>
!     R <pattern_hex> <text_hex>     ->  R <from> <to> <length> <status> <substr_hex>
!     V <pattern_hex> -              ->  V <T|F>
!     L <pattern_hex> -              ->  L <valid T|F> <all_hex> <prefix_hex> <suffix_hex>
!     B <op I|M|R> <pattern_hex> <row_len> <n_rows> <rows_file> <out_file|-> <nthreads>
!           batch/timing mode: rows_file holds n_rows*row_len raw bytes (the storage of a
!           Fortran `character(row_len) :: s(n_rows)`); every row goes through the public
!           operator (per-element compile, as the elemental API does).  Prints
!           `B <seconds> <n_true> <threads>` and, unless out_file is "-", writes one line
!           `<flag 0|1> <from> <to>` per row (from/to only for op R, else 0 0).
program ref_driver
   use, intrinsic :: iso_fortran_env, only: int32, int64, real64, input_unit, output_unit, error_unit
   use :: forgex
   use :: forgex_syntax_tree_graph_m, only: tree_t
   use :: forgex_syntax_tree_optimize_m, only: extract_literal
   !$ use :: omp_lib
   implicit none

   integer, parameter :: LINE_MAX = 1048576
   character(:), allocatable :: line
   integer :: ios

   allocate(character(LINE_MAX) :: line)

   do
      read(input_unit, '(a)', iostat=ios) line
      if (ios /= 0) exit
      if (len_trim(line) == 0) cycle
      call handle(trim(line))
   end do

contains

   subroutine split(str, fields, n)
      character(*), intent(in) :: str
      character(:), allocatable, intent(inout) :: fields(:)
      integer, intent(out) :: n
      integer :: i, b, maxlen
      maxlen = len(str)
      if (allocated(fields)) deallocate(fields)
      allocate(character(maxlen) :: fields(10))
      n = 0
      i = 1
      do while (i <= len(str))
         do while (i <= len(str))
            if (str(i:i) /= ' ') exit
            i = i + 1
         end do
         if (i > len(str)) exit
         b = i
         do while (i <= len(str))
            if (str(i:i) == ' ') exit
            i = i + 1
         end do
         n = n + 1
         if (n > 10) then
            n = 10
            return
         end if
         fields(n) = str(b:i-1)
      end do
   end subroutine split

   function unhex(h) result(s)
      character(*), intent(in) :: h
      character(:), allocatable :: s
      integer :: i, n, v
      if (trim(h) == '-') then
         s = ''
         return
      end if
      n = len_trim(h)/2
      allocate(character(n) :: s)
      do i = 1, n
         read(h(2*i-1:2*i), '(z2)') v
         s(i:i) = achar(v)
      end do
   end function unhex

   function tohex(s) result(h)
      character(*), intent(in) :: s
      character(:), allocatable :: h
      integer :: i
      if (len(s) == 0) then
         h = '-'
         return
      end if
      allocate(character(2*len(s)) :: h)
      do i = 1, len(s)
         write(h(2*i-1:2*i), '(z2.2)') iachar(s(i:i))
      end do
   end function tohex

   character(1) function tf(l)
      logical, intent(in) :: l
      if (l) then
         tf = 'T'
      else
         tf = 'F'
      end if
   end function tf

   subroutine handle(str)
      character(*), intent(in) :: str
      character(:), allocatable :: f(:)
      character(:), allocatable :: pat, txt, sub, all, pre, suf, fac
      integer :: n, from, to, length, status
      logical :: flag
      type(tree_t) :: tree

      call split(str, f, n)
      if (n < 3) then
         write(output_unit, '(a)') 'E bad-line'
         return
      end if

      select case (trim(f(1)))
      case ('I')
         pat = unhex(trim(f(2))); txt = unhex(trim(f(3)))
         flag = pat .in. txt
         write(output_unit, '(a,1x,a)') 'I', tf(flag)
      case ('M')
         pat = unhex(trim(f(2))); txt = unhex(trim(f(3)))
         flag = pat .match. txt
         write(output_unit, '(a,1x,a)') 'M', tf(flag)
      case ('R')
         pat = unhex(trim(f(2))); txt = unhex(trim(f(3)))
         from = -1; to = -1; length = -1; status = -1
         sub = ''
         call regex(pat, txt, sub, length=length, from=from, to=to, status=status)
         write(output_unit, '(a,4(1x,i0),1x,a)') 'R', from, to, length, status, tohex(sub)
      case ('V')
         pat = unhex(trim(f(2)))
         flag = is_valid_regex(pat)
         write(output_unit, '(a,1x,a)') 'V', tf(flag)
      case ('L')
         pat = unhex(trim(f(2)))
         call tree%build(trim(pat))
         if (.not. tree%is_valid) then
            write(output_unit, '(a)') 'L F - - -'
         else
            all = ''; pre = ''; suf = ''; fac = ''
            call extract_literal(tree, all, pre, suf, fac)
            write(output_unit, '(a,1x,a,1x,a,1x,a)') 'L T', tohex(all), tohex(pre), tohex(suf)
         end if
      case ('B')
         call batch(f, n)
      case default
         write(output_unit, '(a)') 'E bad-op'
      end select
      flush(output_unit)
   end subroutine handle

   subroutine batch(f, n)
      character(*), intent(in) :: f(:)
      integer, intent(in) :: n
      character(:), allocatable :: pat, rows
      character(1) :: op
      integer :: row_len, nthreads, u, i, ios, n_true
      integer(int64) :: n_rows, c0, c1, rate
      logical, allocatable :: flags(:)
      integer, allocatable :: froms(:), tos(:)
      character(:), allocatable :: sub

      if (n < 8) then
         write(output_unit, '(a)') 'E bad-batch'
         return
      end if
      op = f(2)(1:1)
      pat = unhex(trim(f(3)))
      read(f(4), *) row_len
      read(f(5), *) n_rows
      read(f(8), *) nthreads
      allocate(character(row_len*n_rows) :: rows)
      open(newunit=u, file=trim(f(6)), access='stream', form='unformatted', status='old', iostat=ios)
      if (ios /= 0) then
         write(output_unit, '(a)') 'E cannot-open-rows'
         return
      end if
      read(u, iostat=ios) rows
      close(u)
      if (ios /= 0) then
         write(output_unit, '(a)') 'E short-rows-file'
         return
      end if
      allocate(flags(n_rows), froms(n_rows), tos(n_rows))
      froms = 0; tos = 0
      !$ call omp_set_num_threads(nthreads)
      call system_clock(c0, rate)
      !$omp parallel do schedule(dynamic, 1) private(sub)
      do i = 1, int(n_rows)
         select case (op)
         case ('I')
            flags(i) = pat .in. rows((i-1)*row_len+1:i*row_len)
         case ('M')
            flags(i) = pat .match. rows((i-1)*row_len+1:i*row_len)
         case default
            sub = ''
            call regex(pat, rows((i-1)*row_len+1:i*row_len), sub, from=froms(i), to=tos(i))
            flags(i) = froms(i) > 0
         end select
      end do
      !$omp end parallel do
      call system_clock(c1)
      n_true = count(flags)
      write(output_unit, '(a,1x,es16.8,1x,i0,1x,i0)') 'B', real(c1-c0, real64)/real(rate, real64), n_true, nthreads
      if (trim(f(7)) /= '-') then
         open(newunit=u, file=trim(f(7)), status='replace', action='write')
         do i = 1, int(n_rows)
            write(u, '(i0,1x,i0,1x,i0)') merge(1, 0, flags(i)), froms(i), tos(i)
         end do
         close(u)
      end if
   end subroutine batch

end program ref_driver
